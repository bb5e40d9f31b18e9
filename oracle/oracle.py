"""ctypes front end of the CPU oracle (oracle/rt_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


class OracleStats(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("shadow", C.c_uint64), ("secondary", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64),
                ("render_ms", C.c_double), ("threads", C.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(force=False):
    src = os.path.join(_HERE, "rt_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "liboracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_scene_new.restype = C.c_void_p
        L.oracle_scene_free.argtypes = [C.c_void_p]
        L.oracle_load_scene.restype = C.c_void_p
        L.oracle_load_scene.argtypes = [C.c_char_p]
        L.oracle_parse_scene.restype = C.c_void_p
        L.oracle_parse_scene.argtypes = [C.c_char_p]
        L.oracle_scene_info.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_scene_output.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.oracle_scene_set_frame.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.oracle_scene_add_shapes.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_scene_add_light.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_scene_export_shapes.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_scene_export_lights.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_scene_export_camera.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_frame_setup.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_render.restype = C.c_int
        L.oracle_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                    C.c_void_p, C.POINTER(OracleStats)]
        L.oracle_render_tiles.restype = C.c_int
        L.oracle_render_tiles.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                          C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(OracleStats)]
        L.oracle_compare.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.oracle_powf.restype = C.c_float
        L.oracle_powf.argtypes = [C.c_float, C.c_float]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class OracleScene:
    """A parsed (or programmatically built) scene held by the oracle."""

    MODE_ALL, MODE_TREE_COLLECT, MODE_TREE_PRUNED = 0, 1, 2

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def load(cls, path):
        h = lib().oracle_load_scene(os.fsencode(path))
        if not h:
            raise ValueError(lib().oracle_last_error().decode())
        return cls(h)

    @classmethod
    def parse(cls, text):
        h = lib().oracle_parse_scene(text.encode())
        if not h:
            raise ValueError(lib().oracle_last_error().decode())
        return cls(h)

    @classmethod
    def empty(cls):
        return cls(lib().oracle_scene_new())

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.oracle_scene_free(self._h)
            self._h = None

    def info(self):
        a = np.zeros(8, dtype=np.uint32)
        lib().oracle_scene_info(self._h, a.ctypes.data)
        keys = ["width", "height", "maxdepth", "nshapes", "nlights", "ntri", "nsph", "npl"]
        return dict(zip(keys, (int(x) for x in a)))

    def output(self):
        buf = C.create_string_buffer(4096)
        lib().oracle_scene_output(self._h, buf, 4096)
        return buf.value.decode()

    def set_frame(self, width, height, maxdepth, camera10, ambient3):
        cam, amb = _f32(camera10), _f32(ambient3)
        lib().oracle_scene_set_frame(self._h, width, height, maxdepth, cam.ctypes.data, amb.ctypes.data)

    def add_shapes(self, kind, geom, mat7):
        geom, mat7 = _f32(geom), _f32(mat7)
        stride = {0: 4, 1: 9, 2: 6}[kind]
        n = geom.size // stride
        per = 1 if mat7.size == 7 * n and n > 1 else 0
        lib().oracle_scene_add_shapes(self._h, kind, n, geom.ctypes.data, mat7.ctypes.data, per)

    def add_light(self, directional, vec3, color3):
        v, c = _f32(vec3), _f32(color3)
        lib().oracle_scene_add_light(self._h, int(directional), v.ctypes.data, c.ctypes.data)

    def export(self):
        """(shapes[n,17], lights[m,7], camera[13]) exactly as parsed; see rt_oracle.cpp."""
        i = self.info()
        shapes = np.zeros((i["nshapes"], 17), dtype=np.float32)
        lights = np.zeros((i["nlights"], 7), dtype=np.float32)
        cam = np.zeros(13, dtype=np.float32)
        lib().oracle_scene_export_shapes(self._h, shapes.ctypes.data)
        lib().oracle_scene_export_lights(self._h, lights.ctypes.data)
        lib().oracle_scene_export_camera(self._h, cam.ctypes.data)
        return shapes, lights, cam

    def frame_setup(self, width=0, height=0):
        i = self.info()
        w, h = (width or i["width"]), (height or i["height"])
        a = np.zeros(16, dtype=np.float32)
        lib().oracle_frame_setup(self._h, w, h, a.ctypes.data)
        return a

    def render(self, width=0, height=0, mode=0, threads=0, rows=None):
        """Returns (pixels[rows, W] uint32 0xFFRRGGBB, stats dict)."""
        i = self.info()
        w, h = (width or i["width"]), (height or i["height"])
        row0, nrows = (0, h) if rows is None else rows
        out = np.zeros((nrows, w), dtype=np.uint32)
        st = OracleStats()
        rc = lib().oracle_render(self._h, w, h, row0, nrows, mode, threads, out.ctypes.data, C.byref(st))
        if rc != 0:
            raise RuntimeError(lib().oracle_last_error().decode())
        return out, st.as_dict()


def _render_tiles(self, shard, nshards, tile_rows=8, width=0, height=0, mode=0, threads=0):
    """Rows of the interleaved tiles owned by `shard`, compact: (pixels[rows, W], stats)."""
    i = self.info()
    w, h = (width or i["width"]), (height or i["height"])
    n = C.c_uint32()
    rc = lib().oracle_render_tiles(self._h, w, h, shard, nshards, tile_rows, mode, threads, None, C.byref(n), None)
    if rc != 0:
        raise RuntimeError(lib().oracle_last_error().decode())
    out = np.zeros((n.value, w), dtype=np.uint32)
    st = OracleStats()
    rc = lib().oracle_render_tiles(self._h, w, h, shard, nshards, tile_rows, mode, threads, out.ctypes.data, C.byref(n), C.byref(st))
    if rc != 0:
        raise RuntimeError(lib().oracle_last_error().decode())
    return out, st.as_dict()


OracleScene.render_tiles = _render_tiles


def compare(a, b):
    """Restatement of the reference's Image::compare plus the parity statistics.

    Returns dict(ref_diff, identical, identical_frac, max_diff, over2) and the diff image.
    Raises ValueError("Images have different dimensions") as the reference returns Err.
    """
    a = np.ascontiguousarray(a, dtype=np.uint32)
    b = np.ascontiguousarray(b, dtype=np.uint32)
    if a.shape != b.shape:
        raise ValueError("Images have different dimensions")
    diff = np.zeros(a.shape, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint64)
    lib().oracle_compare(a.ctypes.data, b.ctypes.data, a.size, diff.ctypes.data, out.ctypes.data)
    n = max(a.size, 1)
    return {"ref_diff": int(out[0]), "identical": int(out[1]), "identical_frac": float(out[1]) / n,
            "max_diff": int(out[2]), "over2": int(out[3])}, diff


def file_to_image(path):
    """imgcomparator/mod.rs:129-142: 8-bit RGB packed as 0x00RRGGBB, row-major."""
    from PIL import Image
    im = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint32)
    return (im[..., 0] << 16) | (im[..., 1] << 8) | im[..., 2]


def save_image(pixels, path):
    """imgcomparator/mod.rs:153-165: drops the top byte, writes 8-bit RGB."""
    from PIL import Image
    p = np.asarray(pixels, dtype=np.uint32)
    rgb = np.stack([(p >> 16) & 0xFF, (p >> 8) & 0xFF, p & 0xFF], axis=-1).astype(np.uint8)
    Image.fromarray(rgb, "RGB").save(path)
