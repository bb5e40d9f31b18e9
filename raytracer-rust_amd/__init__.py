"""MI355X render path of the ray tracer, host side in Python.

The names mirror the reference's Rust API so that callers and tests read the same way:

    ParsedConfigState().load_config_file(path) -> Config     config_builder.rs:119-150
    RayTracer(config)  (RayTracer::new)                      raytracer.rs:29-38
    RayTracer.render() -> Image                              raytracer.rs:40-79
    RayTracer.get_output_path()                              raytracer.rs:81-83
    Image.compare(a, b) -> (count, diff_image)               imgcomparator/mod.rs:54-89
    file_to_image / save_image / extract_rgb                 imgcomparator/mod.rs:100-165

Everything that computes goes through the C ABI of include/rt_amd.h / include/rt_host.h in
librt_amd.so (hand-written gfx950 kernels).  There is no CPU or PyTorch fallback: if the
library is missing or no GPU is usable the calls raise.
"""
import ctypes as C
import importlib.util
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIBRARY") or os.path.join(_HERE, "librt_amd.so")   # the override serves side-by-side builds of the kernels

TILE_ROWS = 8   # rows per interleaved tile when a frame is sharded over GPUs


class RtError(RuntimeError):
    """Err(String) of the reference: the message of the failing call."""


class rt_frame(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("cv", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3),
                ("pixel_w", C.c_float), ("pixel_h", C.c_float), ("half_w", C.c_float), ("half_h", C.c_float),
                ("width", C.c_uint32), ("height", C.c_uint32), ("maxdepth", C.c_uint32), ("ambient", C.c_float * 3)]


_FP = C.POINTER(C.c_float)
_UP = C.POINTER(C.c_uint32)


class rt_scene(C.Structure):
    _fields_ = [("tri_v0", _FP), ("tri_v1", _FP), ("tri_v2", _FP), ("sph_center", _FP), ("sph_radius", _FP),
                ("pl_point", _FP), ("pl_normal", _FP), ("tri_mat", _FP), ("sph_mat", _FP), ("pl_mat", _FP),
                ("tri_order", _UP), ("sph_order", _UP), ("pl_order", _UP),
                ("ntri", C.c_uint32), ("nsph", C.c_uint32), ("npl", C.c_uint32),
                ("lights", _FP), ("nlights", C.c_uint32)]


class rt_stats(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("shadow", C.c_uint64), ("secondary", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64),
                ("box_passes", C.c_uint64), ("shape_passes", C.c_uint64), ("phase_passes", C.c_uint64 * 9),
                ("kernel_ms", C.c_double), ("total_ms", C.c_double)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if k == "phase_passes" else getattr(self, k)) for k, _ in self._fields_}


class rt_bvh_info(C.Structure):
    _fields_ = [("node_pairs", C.c_uint32), ("leaves", C.c_uint32), ("max_depth", C.c_uint32), ("max_leaf", C.c_uint32),
                ("bvh_prims", C.c_uint32), ("list_prims", C.c_uint32), ("build_ms", C.c_double), ("sah_cost", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class rth_config_info(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("maxdepth", C.c_uint32), ("maxverts", C.c_uint32),
                ("ntri", C.c_uint32), ("nsph", C.c_uint32), ("npl", C.c_uint32), ("nlights", C.c_uint32),
                ("camera_position", C.c_float * 3), ("camera_look_at", C.c_float * 3), ("camera_up", C.c_float * 3),
                ("camera_fov", C.c_float), ("ambient", C.c_float * 3)]


# Every symbol include/rt_amd.h and include/rt_host.h declare, with its signature.
_SIGNATURES = {
    "rt_frame_from_camera": (C.c_int, [_FP, _FP, _FP, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, _FP, C.POINTER(rt_frame)]),
    "rt_init": (C.c_int, [C.c_int]),
    "rt_init_multi": (C.c_int, [C.c_int]),
    "rt_device_count": (C.c_int, []),
    "rt_upload_scene": (C.c_int, [C.POINTER(rt_scene)]),
    "rt_upload_scene_ex": (C.c_int, [C.POINTER(rt_scene), C.c_uint32]),
    "rt_get_bvh_info": (C.c_int, [C.POINTER(rt_bvh_info)]),
    "rt_scene_blob": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "rt_copy_scene_blob": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rt_adopt_scene_blob": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rt_upload_scene_blob": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rt_download_scene_blob": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rt_render": (C.c_int, [C.POINTER(rt_frame), C.c_void_p, C.POINTER(rt_stats), C.c_int]),
    "rt_render_frame": (C.c_int, [C.POINTER(rt_frame), C.POINTER(C.c_void_p), C.POINTER(rt_stats), C.c_int]),
    "rt_render_tiles": (C.c_int, [C.POINTER(rt_frame), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(rt_stats), C.c_int]),
    "rt_shard_rows": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rt_assemble_tiles": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "rt_compare_frames": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64)]),
    "rt_compare_frames_noalloc": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "rt_pack_rgb": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "rt_tile_costs_count": (C.c_int, [C.POINTER(rt_frame), C.c_int, C.POINTER(C.c_uint32)]),
    "rt_tile_costs_export": (C.c_int, [C.POINTER(rt_frame), C.c_int, C.c_void_p, C.c_uint32]),
    "rt_tile_costs_import": (C.c_int, [C.POINTER(rt_frame), C.c_int, C.c_void_p, C.c_uint32]),
    "rt_tile_costs_guess": (C.c_int, [C.POINTER(rt_frame), C.c_int, C.c_void_p, C.c_uint32]),
    "rt_select_lane": (C.c_int, [C.c_int]),
    "rt_stream": (C.c_void_p, []),
    "rt_synchronize": (C.c_int, []),
    "rt_wait_event": (C.c_int, [C.c_void_p]),
    "rt_record_event": (C.c_int, [C.c_void_p]),
    "rt_last_error": (C.c_char_p, []),
    "rt_shutdown": (None, []),
    "rth_load_config_file": (C.c_void_p, [C.c_char_p]),
    "rth_parse_config_text": (C.c_void_p, [C.c_char_p]),
    "rth_config_free": (None, [C.c_void_p]),
    "rth_config_get_info": (C.c_int, [C.c_void_p, C.POINTER(rth_config_info)]),
    "rth_config_output_file": (C.c_char_p, [C.c_void_p]),
    "rth_config_set_size": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "rth_config_scene": (C.c_int, [C.c_void_p, C.POINTER(rt_scene)]),
    "rth_config_frame": (C.c_int, [C.c_void_p, C.POINTER(rt_frame)]),
    "rth_image_compare": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64)]),
    "rth_pack_scene": (C.c_int, [C.POINTER(rt_scene), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(rt_bvh_info)]),
    "rth_free": (None, [C.c_void_p]),
    "rth_validate_blob": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rth_save_png": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "rth_save_png_rgb": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "rth_load_png": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rth_scene_cache_write": (C.c_int, [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p]),
    "rth_scene_cache_open": (C.c_void_p, [C.c_char_p]),
    "rth_scene_cache_blob": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "rth_scene_cache_config": (C.c_void_p, [C.c_void_p]),
    "rth_scene_cache_is_fresh": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rth_scene_cache_close": (None, [C.c_void_p]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own copies of the ROCm libraries (libamdhip64.so, librccl.so, the HSA runtime ..., same
    sonames as /opt/rocm's).  Two HIP runtimes in one process do not work (the second one finds no GPU), so when torch is
    installed it is imported before librt_amd.so is loaded, and the library binds to torch's copies by soname.  Buffers and
    streams can then be shared with torch (RCCL collectives on rendered tiles).  Importing torch itself, rather than
    loading its libamdhip64.so and librccl.so by path, matters: with only those two preloaded, a later `import torch`
    brought the rest of its bundle in behind this library's dependencies and the process aborted at exit
    ("double free or corruption") whenever torch was imported after rt_init.  RT_AMD_WITHOUT_TORCH=1 skips all of this
    (a process that never touches torch then runs on /opt/rocm's libraries)."""
    if os.environ.get("RT_AMD_WITHOUT_TORCH"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    try:
        import torch  # noqa: F401
    except Exception:       # noqa: BLE001  (a broken torch must not keep the renderer from loading: fall back to its two libraries by path)
        for name in ("libamdhip64.so", "librccl.so"):
            cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", name)
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """The loaded librt_amd.so.  Raises when it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {_HERE}` (or __graft_entry__.build())")
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here = header and library disagree
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise RtError(lib().rt_last_error().decode(errors="replace"))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(_FP) if a.size else C.cast(None, _FP)


# ------------------------------------------------------------------------------------ scene model
class Config:
    """Config of config_builder.rs:14-24 (opaque handle into the C++ twin)."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rth_config_free(self._h)
            self._h = None

    def _info(self):
        i = rth_config_info()
        _check(lib().rth_config_get_info(self._h, C.byref(i)))
        return i

    width = property(lambda self: self._info().width)
    height = property(lambda self: self._info().height)
    maxdepth = property(lambda self: self._info().maxdepth)
    maxverts = property(lambda self: self._info().maxverts)
    ambient = property(lambda self: tuple(self._info().ambient))

    @property
    def output_file(self):
        return lib().rth_config_output_file(self._h).decode()

    @property
    def camera(self):
        i = self._info()
        return {"position": tuple(i.camera_position), "look_at": tuple(i.camera_look_at), "up": tuple(i.camera_up), "fov": i.camera_fov}

    def counts(self):
        i = self._info()
        return {"ntri": i.ntri, "nsph": i.nsph, "npl": i.npl, "nlights": i.nlights}

    def set_size(self, width, height):
        """What editing the `size` line does (the 3840x2160 dragon configuration)."""
        _check(lib().rth_config_set_size(self._h, width, height))

    def scene(self):
        s = rt_scene()
        _check(lib().rth_config_scene(self._h, C.byref(s)))
        s._owner = self          # the arrays live inside the config
        return s

    def frame(self):
        f = rt_frame()
        _check(lib().rth_config_frame(self._h, C.byref(f)))
        return f


class ParsedConfigState:
    """ParsedConfigState of config_builder.rs:111-150."""

    def load_config_file(self, file_path):
        h = lib().rth_load_config_file(os.fsencode(file_path))
        if not h:
            raise RtError(lib().rt_last_error().decode(errors="replace"))
        return Config(h)

    def parse_text(self, text):
        h = lib().rth_parse_config_text(text.encode())
        if not h:
            raise RtError(lib().rt_last_error().decode(errors="replace"))
        return Config(h)


class SceneArrays:
    """A scene given directly as arrays (synthetic meshes too large for a text file).

    Keeps the numpy buffers alive for as long as the rt_scene that points into them.
    """

    def __init__(self, tri=None, tri_mat=None, spheres=None, sph_mat=None, planes=None, pl_mat=None, lights=None,
                 tri_order=None, sph_order=None, pl_order=None):
        z = np.zeros((0,), dtype=np.float32)
        tri = _f32(tri).reshape(-1, 9) if tri is not None else z.reshape(0, 9)
        self.v0, self.v1, self.v2 = _f32(tri[:, 0:3]), _f32(tri[:, 3:6]), _f32(tri[:, 6:9])
        sph = _f32(spheres).reshape(-1, 4) if spheres is not None else z.reshape(0, 4)
        self.sc, self.sr = _f32(sph[:, 0:3]), _f32(sph[:, 3])
        pl = _f32(planes).reshape(-1, 6) if planes is not None else z.reshape(0, 6)
        self.pp, self.pn = _f32(pl[:, 0:3]), _f32(pl[:, 3:6])

        def mats(m, n):
            m = _f32(m if m is not None else z).reshape(-1, 7) if n else z.reshape(0, 7)
            if n and m.shape[0] == 1:
                m = np.repeat(m, n, axis=0)
            assert m.shape[0] == n, "one material (7 floats) per shape, or a single one for all"
            return _f32(m)

        self.tm, self.sm, self.pm = mats(tri_mat, len(tri)), mats(sph_mat, len(sph)), mats(pl_mat, len(pl))
        self.lights = _f32(lights).reshape(-1, 7) if lights is not None else z.reshape(0, 7)
        self.orders = [None if o is None else np.ascontiguousarray(o, dtype=np.uint32) for o in (tri_order, sph_order, pl_order)]
        s = rt_scene()
        s.tri_v0, s.tri_v1, s.tri_v2, s.tri_mat = _fp(self.v0), _fp(self.v1), _fp(self.v2), _fp(self.tm)
        s.sph_center, s.sph_radius, s.sph_mat = _fp(self.sc), _fp(self.sr), _fp(self.sm)
        s.pl_point, s.pl_normal, s.pl_mat = _fp(self.pp), _fp(self.pn), _fp(self.pm)
        for name, o in zip(("tri_order", "sph_order", "pl_order"), self.orders):
            setattr(s, name, o.ctypes.data_as(_UP) if o is not None else C.cast(None, _UP))
        s.ntri, s.nsph, s.npl = len(tri), len(sph), len(pl)
        s.lights, s.nlights = _fp(self.lights), len(self.lights)
        # The struct only holds raw pointers: tie the buffers' lifetime to it as well, so that
        # `SceneArrays(...).c` stays valid after the wrapper object is gone.
        s._buffers = (self.v0, self.v1, self.v2, self.sc, self.sr, self.pp, self.pn, self.tm, self.sm, self.pm, self.lights, self.orders)
        self.c = s


def frame_from_camera(position, look_at, up, fov, width, height, maxdepth, ambient):
    p, l, u, a = _f32(position), _f32(look_at), _f32(up), _f32(ambient)
    f = rt_frame()
    _check(lib().rt_frame_from_camera(_fp(p), _fp(l), _fp(u), float(fov), width, height, maxdepth, _fp(a), C.byref(f)))
    return f


def pack_scene_host(scene):
    """Host-only build of the device blob: (bytes as np.uint8 array, bvh info dict)."""
    blob, n, info = C.c_void_p(), C.c_size_t(), rt_bvh_info()
    _check(lib().rth_pack_scene(C.byref(scene), C.byref(blob), C.byref(n), C.byref(info)))
    try:
        out = np.ctypeslib.as_array(C.cast(blob, C.POINTER(C.c_uint8)), shape=(n.value,)).copy()
    finally:
        lib().rth_free(blob)
    return out, info.as_dict()


def validate_blob(blob):
    """Raises RtError unless `blob` (np.uint8 array) is a well-formed scene blob."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    _check(lib().rth_validate_blob(blob.ctypes.data, blob.size))


# ------------------------------------------------------------------------------------ images
class Image:
    """Image of imgcomparator/mod.rs:21-40: width, height, packed 0x??RRGGBB pixels."""

    def __init__(self, width, height, data):
        self.width, self.height = int(width), int(height)
        self.data = np.ascontiguousarray(data, dtype=np.uint32).reshape(-1)

    def pixels(self):
        return self.data.reshape(self.height, self.width)

    @staticmethod
    def compare(img1, img2):
        """(number of pixels with a channel differing by more than 1, difference image)."""
        n = max(img1.data.size, 1)
        diff = np.zeros(n, dtype=np.uint32)
        count = C.c_uint64()
        _check(lib().rth_image_compare(img1.data.ctypes.data, img1.width, img1.height, img2.data.ctypes.data, img2.width, img2.height,
                                       diff.ctypes.data, C.byref(count)))
        return int(count.value), Image(img1.width, img1.height, diff[:img1.data.size])


def extract_rgb(pixel):
    return (pixel >> 16) & 0xFF, (pixel >> 8) & 0xFF, pixel & 0xFF


def file_to_image(path):
    """file_to_image of imgcomparator/mod.rs:128-148: a PNG as 0x00RRGGBB pixels."""
    p, w, h = C.c_void_p(), C.c_uint32(), C.c_uint32()
    _check(lib().rth_load_png(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h)))
    try:
        data = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(w.value * h.value,)).copy()
    finally:
        lib().rth_free(p)
    return Image(w.value, h.value, data)


def save_image(img, path):
    """save_image of imgcomparator/mod.rs:150-166: 8-bit RGB PNG."""
    _check(lib().rth_save_png(os.fsencode(path), img.data.ctypes.data, img.width, img.height))


def save_image_rgb(rgb, width, height, path):
    """The same from R,G,B bytes (what pack_rgb leaves in device memory, copied to the host)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    assert rgb.size == 3 * width * height
    _check(lib().rth_save_png_rgb(os.fsencode(path), rgb.ctypes.data, width, height))


# ------------------------------------------------------------------------------------ renderer
_bound_device = None


def init(device=0):
    """Binds this process to one GPU (one process per GPU)."""
    global _bound_device
    _check(lib().rt_init(int(device)))
    _bound_device = int(device)


def init_multi(ngpus):
    """Binds this process to the first `ngpus` GPUs of the node: scenes are replicated and frames sharded inside the library."""
    global _bound_device
    _check(lib().rt_init_multi(int(ngpus)))
    _bound_device = 0


def device_count():
    return int(lib().rt_device_count())


def shutdown():
    global _bound_device, _resident_token
    if _lib is not None:
        _lib.rt_shutdown()
    _bound_device = None
    _resident_token += 1            # no scene is resident any more: every RayTracer uploads its own again after the next init


BUILD_ON_GPU = 1
_resident_token = 0

# The library keeps ONE scene resident.  Every upload takes a new token; a RayTracer whose
# token is no longer the current one makes its own scene resident again before it renders (the reference's RayTracer
# owns its Config and Bvh, so holding several at once is valid there).


def _new_resident_token():
    global _resident_token
    _resident_token += 1
    return _resident_token


def upload_scene(scene, gpu_build=False):
    """RayTracer::new: tree build (split search on the host, or on the GPU with gpu_build=True) + upload."""
    try:
        _check(lib().rt_upload_scene_ex(C.byref(scene), BUILD_ON_GPU if gpu_build else 0))
    finally:
        _new_resident_token()       # also when it failed: the scene that was resident may be gone
    info = rt_bvh_info()
    _check(lib().rt_get_bvh_info(C.byref(info)))
    return info.as_dict()


def scene_blob():
    p, n = C.c_void_p(), C.c_size_t()
    _check(lib().rt_scene_blob(C.byref(p), C.byref(n)))
    return p.value, n.value


def copy_scene_blob(device_dst, nbytes):
    _check(lib().rt_copy_scene_blob(C.c_void_p(device_dst), nbytes))


def adopt_scene_blob(device_ptr, nbytes):
    try:
        _check(lib().rt_adopt_scene_blob(C.c_void_p(device_ptr), nbytes))
    finally:
        _new_resident_token()


def upload_scene_blob(blob):
    """A packed scene (np.uint8 array from pack_scene_host, or a SceneCache) -> resident scene; no tree build."""
    if isinstance(blob, SceneCache):
        ptr, n = blob.blob_pointer()
    else:
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        ptr, n = blob.ctypes.data, blob.size
    try:
        _check(lib().rt_upload_scene_blob(C.c_void_p(ptr), n))
    finally:
        _new_resident_token()
    info = rt_bvh_info()
    _check(lib().rt_get_bvh_info(C.byref(info)))
    return info.as_dict()


def download_scene_blob():
    """The resident scene blob as a np.uint8 array."""
    _, n = scene_blob()
    out = np.empty(n, dtype=np.uint8)
    _check(lib().rt_download_scene_blob(out.ctypes.data, n))
    return out


class SceneCache:
    """A binary scene cache file, memory-mapped (see rt_host.h)."""

    def __init__(self, path):
        h = lib().rth_scene_cache_open(os.fsencode(path))
        if not h:
            raise RtError(lib().rt_last_error().decode(errors="replace"))
        self._h = h

    @staticmethod
    def write(path, config, blob, source_path=None):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        _check(lib().rth_scene_cache_write(os.fsencode(path), config._h, blob.ctypes.data, blob.size,
                                           os.fsencode(source_path) if source_path else None))

    def blob_pointer(self):
        n = C.c_size_t()
        p = lib().rth_scene_cache_blob(self._h, C.byref(n))
        if not p:
            raise RtError(lib().rt_last_error().decode(errors="replace"))
        return p, n.value

    def blob(self):
        p, n = self.blob_pointer()
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,)).copy()

    def config(self):
        h = lib().rth_scene_cache_config(self._h)
        if not h:
            raise RtError(lib().rt_last_error().decode(errors="replace"))
        return Config(h)

    def is_fresh(self, source_path):
        return bool(lib().rth_scene_cache_is_fresh(self._h, os.fsencode(source_path)))

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rth_scene_cache_close(self._h)
        self._h = None

    __del__ = close


def shard_rows(height, shard, nshards, tile_rows=TILE_ROWS):
    return int(lib().rt_shard_rows(height, shard, nshards, tile_rows))


def render(frame, count_work=False, out=None):
    """Whole frame into host memory: (pixels[H, W] uint32, stats dict).  `out`: a C-contiguous uint32 array of that shape to
    render into (a caller that renders frame after frame keeps its buffer, as it would keep a Vec<u32>)."""
    if out is None:
        out = np.empty((frame.height, frame.width), dtype=np.uint32)
    elif out.dtype != np.uint32 or out.shape != (frame.height, frame.width) or not out.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous uint32 array of shape (height, width)")
    st = rt_stats()
    _check(lib().rt_render(C.byref(frame), out.ctypes.data, C.byref(st), int(bool(count_work))))
    return out, st.as_dict()


def render_frame(frame, want_stats=False, count_work=False):
    """Whole frame left in device memory of the first GPU: (device pointer, stats dict or None).  Without stats the call
    only enqueues on the current lane (synchronize() waits)."""
    p, st = C.c_void_p(), rt_stats()
    _check(lib().rt_render_frame(C.byref(frame), C.byref(p), C.byref(st) if (want_stats or count_work) else None, int(bool(count_work))))
    return p.value, (st.as_dict() if (want_stats or count_work) else None)


def render_tiles(frame, shard, nshards, device_ptr, tile_rows=TILE_ROWS, want_stats=False, count_work=False):
    """One shard's row tiles into a device buffer (compact rows)."""
    st = rt_stats()
    _check(lib().rt_render_tiles(C.byref(frame), shard, nshards, tile_rows, C.c_void_p(device_ptr),
                                 C.byref(st) if (want_stats or count_work) else None, int(bool(count_work))))
    return st.as_dict() if (want_stats or count_work) else None


def assemble_tiles(gathered_ptr, stride_rows, nshards, width, height, frame_ptr, tile_rows=TILE_ROWS):
    _check(lib().rt_assemble_tiles(C.c_void_p(gathered_ptr), stride_rows, nshards, tile_rows, width, height, C.c_void_p(frame_ptr)))


def compare_frames(ptr_a, size_a, ptr_b, size_b, diff_ptr=None):
    """Image::compare on two frames in device memory; sizes are (width, height).  Returns the count."""
    count = C.c_uint64()
    _check(lib().rt_compare_frames(C.c_void_p(ptr_a), size_a[0], size_a[1], C.c_void_p(ptr_b), size_b[0], size_b[1],
                                   C.c_void_p(diff_ptr) if diff_ptr else None, C.byref(count)))
    return int(count.value)


def compare_frames_noalloc(ptr_a, size_a, ptr_b, size_b, count_ptr, diff_ptr=None):
    """compare_frames without any allocation: the count goes to device memory at count_ptr (uint64), asynchronously on stream()."""
    _check(lib().rt_compare_frames_noalloc(C.c_void_p(ptr_a), size_a[0], size_a[1], C.c_void_p(ptr_b), size_b[0], size_b[1],
                                           C.c_void_p(diff_ptr) if diff_ptr else None, C.c_void_p(count_ptr)))


def pack_rgb(pixels_ptr, width, height, rgb_ptr):
    """0x??RRGGBB words -> R,G,B bytes, device to device, asynchronous on stream()."""
    _check(lib().rt_pack_rgb(C.c_void_p(pixels_ptr), width, height, C.c_void_p(rgb_ptr)))


LANES = 4          # RT_LANES of rt_amd.h


def select_lane(lane):
    """Chooses which of the LANES lanes (stream + work buffers each) the following calls use."""
    _check(lib().rt_select_lane(int(lane)))


def synchronize():
    _check(lib().rt_synchronize())


def stream():
    return lib().rt_stream()


def wait_event(hip_event):
    """Work enqueued on the current lane from now on runs after `hip_event` (a hipEvent_t handle, e.g. torch.cuda.Event.cuda_event)."""
    _check(lib().rt_wait_event(C.c_void_p(hip_event)))


def record_event(hip_event):
    """Records `hip_event` behind the work enqueued on the current lane so far."""
    _check(lib().rt_record_event(C.c_void_p(hip_event)))


def after_torch(torch_stream=None):
    """Orders the current lane behind what has been enqueued on a torch stream (default: torch's current stream) so far:
    tensors filled there may be handed to the library right away, without a host-side wait."""
    import torch
    ev = torch.cuda.Event()
    ev.record(torch_stream if torch_stream is not None else torch.cuda.current_stream())
    wait_event(ev.cuda_event)
    return ev           # (keep it alive until the lane has passed it; rt_synchronize or any later wait does)


def before_torch(torch_stream=None):
    """The inverse: a torch stream (default: the current one) waits for what the current lane has been given so far."""
    import torch
    target = torch_stream if torch_stream is not None else torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(target)                   # torch creates the HIP event on its first record; the library then records it anew
    record_event(ev.cuda_event)
    target.wait_event(ev)
    return ev


def tile_costs_count(frame, gpu=0):
    """Number of tile regions whose cost a launch of `frame` on GPU `gpu` measures (0: too small a frame to be ordered)."""
    n = C.c_uint32(0)
    _check(lib().rt_tile_costs_count(C.byref(frame), int(gpu), C.byref(n)))
    return n.value


def tile_costs_export(frame, gpu=0):
    """Costs measured by the last launch of `frame` on the current lane (uint32 per region)."""
    costs = np.zeros(tile_costs_count(frame, gpu), dtype=np.uint32)
    _check(lib().rt_tile_costs_export(C.byref(frame), int(gpu), costs.ctypes.data, len(costs)))
    return costs


def tile_costs_guess(frame, gpu=0):
    """What the first launch of `frame` since the upload is ordered by: shapes of the tree seen in and around every region,
    directly and in the specular planes (uint32 per region)."""
    costs = np.zeros(tile_costs_count(frame, gpu), dtype=np.uint32)
    _check(lib().rt_tile_costs_guess(C.byref(frame), int(gpu), costs.ctypes.data, len(costs)))
    return costs


def tile_costs_import(frame, costs, gpu=0):
    """Orders the next launch of `frame` on the current lane by costs saved from an earlier one."""
    costs = np.ascontiguousarray(costs, dtype=np.uint32)
    _check(lib().rt_tile_costs_import(C.byref(frame), int(gpu), costs.ctypes.data, len(costs)))


TILE_FILE_MAGIC = 0x31545452      # "RTT1"


def save_tile_costs(path, frame):
    """Writes the costs of the last launch of `frame` on every GPU the library drives to `path`:
    [magic, bytes of the frame, GPUs, then per GPU its count and its costs], all uint32."""
    words = [np.uint32([TILE_FILE_MAGIC, C.sizeof(frame)]), np.frombuffer(bytes(frame), dtype=np.uint8).astype(np.uint32), np.uint32([device_count()])]
    for gpu in range(device_count()):
        costs = tile_costs_export(frame, gpu)
        words += [np.uint32([len(costs)]), costs]
    tmp = f"{path}.{os.getpid()}.tmp"
    np.concatenate(words).tofile(tmp)
    os.replace(tmp, path)


def load_tile_costs(path, frame):
    """Hands the costs in `path` to the library if they were measured for this very frame on as many GPUs;
    returns whether it did.  Anything else in the file is ignored: the costs only order the work."""
    try:
        w = np.fromfile(path, dtype=np.uint32)
    except OSError:
        return False
    nb = C.sizeof(frame)
    if len(w) < 3 + nb or w[0] != TILE_FILE_MAGIC or w[1] != nb or w[2 + nb] != device_count():
        return False
    if bytes(w[2:2 + nb].astype(np.uint8)) != bytes(frame):
        return False
    at, per_gpu = 3 + nb, []
    for gpu in range(device_count()):
        if at >= len(w) or at + 1 + int(w[at]) > len(w) or int(w[at]) != tile_costs_count(frame, gpu):
            return False
        per_gpu.append(w[at + 1:at + 1 + int(w[at])])
        at += 1 + int(w[at])
    for gpu, costs in enumerate(per_gpu):
        tile_costs_import(frame, costs, gpu)
    return True


GPU_BUILD_FROM = 200_000      # shapes from which from_scene_file builds the tree on the GPU unless told otherwise

class RayTracer:
    """RayTracer of raytracer.rs:16-83 on one MI355X."""

    def __init__(self, config, device=None, gpu_build=False):
        self._device = 0 if device is None else device
        if _bound_device is None:
            init(self._device)
        self.config = config
        self._scene = config.scene() if isinstance(config, Config) else config.c
        self._gpu_build, self._cache_path = gpu_build, None
        self.bvh_info = upload_scene(self._scene, gpu_build=gpu_build)          # RayTracer::new builds the tree (raytracer.rs:32)
        self._token = _resident_token
        self.last_stats = None
        self._tiles_path, self._tiles_token = None, None

    @classmethod
    def new(cls, config):
        return cls(config)

    @classmethod
    def from_scene_file(cls, path, cache_path=None, device=None, gpu_build=None):
        """main.rs:9-11 with a binary cache beside the scene file: the first start parses the text,
        builds the tree and writes `cache_path` (default: path + ".rtcache"); later starts map the
        cache and upload it, as long as the scene file is unchanged.  gpu_build: where the tree is built
        on the first start (None: on the GPU from GPU_BUILD_FROM shapes on, where the host builder takes
        longer than everything else put together; the frame does not depend on it).
        The costs of the tile regions measured by the first frame go to cache_path + ".tiles", and the first frame of a
        later start is handed out in that order (a one-frame run like main.rs:11-15 otherwise never sees an ordered launch)."""
        if _bound_device is None:
            init(0 if device is None else device)
        cache_path = cache_path or (os.fspath(path) + ".rtcache")
        self = cls.__new__(cls)
        self._device = 0 if device is None else device
        self.last_stats = None
        self._cache_path, self._gpu_build = cache_path, False
        self._tiles_path, self._tiles_token = cache_path + ".tiles", None
        try:
            cache = SceneCache(cache_path)
        except RtError:
            cache = None
        if cache is not None and cache.is_fresh(path):
            self.config = cache.config()
            self._scene = None
            self.bvh_info = upload_scene_blob(cache)
            self._token = _resident_token
            self.from_cache = True
            cache.close()
            return self
        if cache is not None:
            cache.close()
        self.config = ParsedConfigState().load_config_file(path)
        self._scene = self.config.scene()
        if gpu_build is None:
            gpu_build = self._scene.ntri + self._scene.nsph >= GPU_BUILD_FROM
        self._gpu_build = gpu_build
        self.bvh_info = upload_scene(self._scene, gpu_build=gpu_build)
        self._token = _resident_token
        SceneCache.write(cache_path, self.config, download_scene_blob(), source_path=path)
        self.from_cache = False
        return self

    def _make_resident(self):
        """Uploads this tracer's scene again if it is no longer the resident one: another RayTracer has replaced it, an
        upload has failed in between, or the library has been shut down (then it is bound to a device again first)."""
        if self._token == _resident_token:
            return
        if _bound_device is None:
            init(self._device)
        if self._scene is not None:
            upload_scene(self._scene, gpu_build=self._gpu_build)
        else:
            cache = SceneCache(self._cache_path)
            try:
                upload_scene_blob(cache)
            finally:
                cache.close()
        self._token = _resident_token

    def render(self, count_work=False):
        self._make_resident()
        frame = self.config.frame()
        fresh = self._tiles_path is not None and self._tiles_token != _resident_token     # first frame since this scene became resident
        if fresh:
            load_tile_costs(self._tiles_path, frame)
        pixels, self.last_stats = render(frame, count_work=count_work)
        if fresh:                                              # (also when the order came from the file: every start refines it)
            self._tiles_token = _resident_token
            if tile_costs_count(frame) > 0:
                try:
                    save_tile_costs(self._tiles_path, frame)
                except OSError:
                    pass                                       # a read-only directory: the next start orders nothing, as this one
        return Image(pixels.shape[1], pixels.shape[0], pixels)

    def get_output_path(self):
        return self.config.output_file
