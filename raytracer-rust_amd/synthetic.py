"""Procedural height-field scene for the large-mesh configuration (about 10 M triangles at
7680x4320), generated in memory: a .scene text of this size would be around half a gigabyte.

Definition (everything in binary32 unless noted):
  vertices  (i, j), 0 <= i, j < n:  x = -50 + 100 * i / (n - 1),  y = -50 + 100 * j / (n - 1)   (computed in float64, rounded once)
            z = 4 sin(0.11 x) cos(0.13 y) + 1.5 sin(0.37 x + 0.29 y) + 0.25 (hash32(i, j, seed) / 2^32 - 0.5)   (float64, rounded once)
  hash32    h = i * 0x9E3779B1 ^ j * 0x85EBCA77 ^ seed;  h ^= h >> 15;  h *= 0x2C1B3C6D;  h ^= h >> 12;  h *= 0x297A2D39;  h ^= h >> 15   (mod 2^32)
  triangles two per grid cell, in row-major cell order: (v00, v10, v11) then (v00, v11, v01), v_ab = vertex (i + a, j + b)
            material diffuse (.5, .4, .3), specular (.3, .3, .3), shininess 20
  plane     point (0, 0, -10), normal (0, 0, 1), diffuse (.2, .2, .2), specular (.5, .5, .5), shininess 20 (after the triangles in file order)
  lights    4 point lights at (+-40, +-40, 60), colour (.25, .25, .25), order (40,40), (40,-40), (-40,40), (-40,-40)
  frame     camera 60 -80 70 -> 0 0 0, up 0 0 1, fov 45; ambient 0; maxdepth 3

n = 2237 gives 2 * 2236^2 = 9 998 592 triangles.
"""
import numpy as np

DEFAULT_N = 2237
SEED = 0x5EED
TRI_MATERIAL = (0.5, 0.4, 0.3, 0.3, 0.3, 0.3, 20.0)
PLANE = (0.0, 0.0, -10.0, 0.0, 0.0, 1.0)
PLANE_MATERIAL = (0.2, 0.2, 0.2, 0.5, 0.5, 0.5, 20.0)
LIGHTS = [(0.0, sx * 40.0, sy * 40.0, 60.0, 0.25, 0.25, 0.25) for sx in (1, -1) for sy in (1, -1)]
CAMERA = dict(position=(60.0, -80.0, 70.0), look_at=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0), fov=45.0)
AMBIENT = (0.0, 0.0, 0.0)
MAXDEPTH = 3


def hash32(i, j, seed=SEED):
    with np.errstate(over="ignore"):
        h = (i.astype(np.uint32) * np.uint32(0x9E3779B1)) ^ (j.astype(np.uint32) * np.uint32(0x85EBCA77)) ^ np.uint32(seed)
        h ^= h >> np.uint32(15)
        h *= np.uint32(0x2C1B3C6D)
        h ^= h >> np.uint32(12)
        h *= np.uint32(0x297A2D39)
        h ^= h >> np.uint32(15)
    return h


def heightfield_vertices(n=DEFAULT_N, seed=SEED):
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    x = -50.0 + 100.0 * i / (n - 1)
    y = -50.0 + 100.0 * j / (n - 1)
    noise = hash32(i, j, seed).astype(np.float64) / 4294967296.0 - 0.5
    z = 4.0 * np.sin(0.11 * x) * np.cos(0.13 * y) + 1.5 * np.sin(0.37 * x + 0.29 * y) + 0.25 * noise
    return np.stack([x, y, z], axis=-1).astype(np.float32)           # [n, n, 3]


def heightfield_triangles(n=DEFAULT_N, seed=SEED):
    """float32 array [2 (n-1)^2, 9]: v0, v1, v2 of every triangle in file order."""
    v = heightfield_vertices(n, seed)
    v00, v10, v11, v01 = v[:-1, :-1], v[1:, :-1], v[1:, 1:], v[:-1, 1:]
    a = np.concatenate([v00, v10, v11], axis=-1)                      # [n-1, n-1, 9]
    b = np.concatenate([v00, v11, v01], axis=-1)
    return np.stack([a, b], axis=2).reshape(-1, 9)


def product_scene(rt, n=DEFAULT_N, seed=SEED):
    """rt.SceneArrays for the product library."""
    tri = heightfield_triangles(n, seed)
    return rt.SceneArrays(tri=tri, tri_mat=TRI_MATERIAL, planes=[PLANE], pl_mat=PLANE_MATERIAL, lights=LIGHTS)


def product_frame(rt, width, height, maxdepth=MAXDEPTH):
    return rt.frame_from_camera(CAMERA["position"], CAMERA["look_at"], CAMERA["up"], CAMERA["fov"], width, height, maxdepth, AMBIENT)


def oracle_scene(oracle, width, height, n=DEFAULT_N, seed=SEED, maxdepth=MAXDEPTH):
    """The same scene handed to the CPU oracle (tests only)."""
    sc = oracle.OracleScene.empty()
    cam = list(CAMERA["position"]) + list(CAMERA["look_at"]) + list(CAMERA["up"]) + [CAMERA["fov"]]
    sc.set_frame(width, height, maxdepth, cam, AMBIENT)
    sc.add_shapes(1, heightfield_triangles(n, seed), TRI_MATERIAL)
    sc.add_shapes(2, PLANE, PLANE_MATERIAL)
    for l in LIGHTS:
        sc.add_light(l[0] != 0.0, l[1:4], l[4:7])
    return sc
