"""Seeded synthetic RGB-D streams for the fusion path (measurement + test inputs).

SURVEY.md s.8(d) defines the scenes.  The reference consumes TUM-style datasets through
Tools/DatasetWrapper.hpp:55-263; none is available offline, so the harness generates:

* S-room: camera inside an axis-aligned 4 x 3 x 4 m box, orbiting; analytic ray/box z-depth.
* S-wall: fronto-parallel plane (known-answer scene).

Holes (depth 0) are mandatory: ChunkManager::findCubeCornerByMat back-projects only
``depth + 0.2`` (Structure/ChunkManager.h:331), so a hole-free fronto-parallel wall selects no
chunks at all.

Everything is numpy; nothing here touches the GPU or the oracle.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np


@dataclasses.dataclass(frozen=True)
class Camera:
    width: int = 640
    height: int = 480
    fx: float = 525.0
    fy: float = 525.0
    cx: float = 319.5
    cy: float = 239.5
    near: float = 0.01
    far: float = 5.0

    @staticmethod
    def hires() -> "Camera":
        return Camera(1280, 960, 1050.0, 1050.0, 639.5, 479.5, 0.01, 5.0)


def pose_identity() -> np.ndarray:
    p = np.zeros((3, 4), np.float32)
    p[0, 0] = p[1, 1] = p[2, 2] = 1.0
    return p


def pose_yaw(yaw: float, t=(0.0, 0.0, 0.0)) -> np.ndarray:
    """Camera-to-world [R|t], rotation about +y."""
    c, s = math.cos(yaw), math.sin(yaw)
    p = np.array([[c, 0.0, s, t[0]], [0.0, 1.0, 0.0, t[1]], [-s, 0.0, c, t[2]]], np.float64)
    return p.astype(np.float32)


def pose_euler(yaw: float, pitch: float, roll: float, t=(0.0, 0.0, 0.0)) -> np.ndarray:
    cy, sy = math.cos(yaw), math.sin(yaw)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cr, sr = math.cos(roll), math.sin(roll)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], np.float64)
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]], np.float64)
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]], np.float64)
    R = Ry @ Rx @ Rz
    p = np.concatenate([R, np.asarray(t, np.float64).reshape(3, 1)], axis=1)
    return p.astype(np.float32)


def _rays(cam: Camera):
    u = (np.arange(cam.width, dtype=np.float64) - cam.cx) / cam.fx
    v = (np.arange(cam.height, dtype=np.float64) - cam.cy) / cam.fy
    uu, vv = np.meshgrid(u, v)
    return np.stack([uu, vv, np.ones_like(uu)], axis=-1)  # [H, W, 3], z == 1


def _holes(shape, frac: float, seed: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.random(shape) < frac


def _hash_colour(world: np.ndarray, seed: int) -> np.ndarray:
    """RGBA8 = integer hash of the world position quantised to 2 cm; A = 1."""
    q = np.floor(world / 0.02).astype(np.int64)
    h = (q[..., 0] * 73856093) ^ (q[..., 1] * 19349663) ^ (q[..., 2] * 83492791) ^ (seed * 2654435761)
    h = (h ^ (h >> 13)) * 0x5BD1E995
    h = h ^ (h >> 15)
    rgba = np.empty(world.shape[:-1] + (4,), np.uint8)
    rgba[..., 0] = (h & 0xFF).astype(np.uint8)
    rgba[..., 1] = ((h >> 8) & 0xFF).astype(np.uint8)
    rgba[..., 2] = ((h >> 16) & 0xFF).astype(np.uint8)
    rgba[..., 3] = 1
    return rgba


def room_frame(k: int, cam: Camera = Camera(), n_orbit: int = 200, hole_frac: float = 0.02,
               half=(2.0, 1.5, 2.0), radius: float = 0.3, with_quality: bool = True, wobble: float = 0.0):
    """Frame k of S-room.  Returns (depth f32[H,W], rgba u8[H,W,4], quality f32[H,W], pose f32[3,4]).
    wobble > 0 (radians) adds a hand-held pitch / roll on top of the yaw orbit: general rotation matrices
    (the plain orbit's have four exact zeros, which hides every summation-order question)."""
    yaw = 2.0 * math.pi * k / n_orbit
    t = (radius * math.sin(yaw), 0.0, radius * math.cos(yaw))
    if wobble:
        pose = pose_euler(yaw, wobble * math.sin(7.0 * yaw + 0.3), wobble * math.cos(5.0 * yaw + 1.1), t)
    else:
        pose = pose_yaw(yaw, t)
    R = pose[:, :3].astype(np.float64)
    o = pose[:, 3].astype(np.float64)
    d_cam = _rays(cam)
    d_w = d_cam @ R.T
    hb = np.asarray(half, np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        s_hi = (hb - o) / d_w
        s_lo = (-hb - o) / d_w
    s_exit = np.where(d_w > 0, s_hi, np.where(d_w < 0, s_lo, np.inf))
    s = s_exit.min(axis=-1)  # camera is inside the box: first exit; z-depth == s because d_cam.z == 1
    world = o + d_w * s[..., None]
    depth = s.astype(np.float32)
    holes = _holes(depth.shape, hole_frac, 1234 + k)
    depth[holes] = 0.0
    rgba = _hash_colour(world, 99)
    quality = None
    if with_quality:
        rng = np.random.Generator(np.random.PCG64(7 + k))
        quality = rng.random(depth.shape, dtype=np.float32)
    return depth, rgba, quality, pose


def wall_frame(z: float = 1.5, cam: Camera = Camera(), pose: np.ndarray | None = None,
               hole_stride: int = 53, rgba_value=(200, 100, 50, 1), quality_value: float = 0.25,
               seed: int = 0):
    """S-wall: plane at camera-frame depth z (constant z-depth image), every hole_stride-th pixel a hole."""
    if pose is None:
        pose = pose_identity()
    depth = np.full((cam.height, cam.width), z, np.float32)
    if hole_stride:
        flat = depth.reshape(-1)
        flat[seed % hole_stride::hole_stride] = 0.0
    rgba = np.empty((cam.height, cam.width, 4), np.uint8)
    rgba[...] = np.asarray(rgba_value, np.uint8)
    quality = np.full((cam.height, cam.width), quality_value, np.float32)
    return depth, rgba, quality, pose


def wall_mesh_for_chunks(ids, res, z0: float, seed: int = 5):
    """Synthetic per-chunk meshes for the atlas stage: for every chunk whose z-extent contains the
    plane z = z0, a small cloud of vertices on that plane inside the chunk (the reference gets
    them from marching cubes, Structure/ChunkManager.cpp:595-1002 -- next-stage scope).
    Returns (chunk ids [m,3], vertex offsets [m+1], verts [nv,3] f32, colors [nv,3] f32 in [0,1])."""
    ids = np.asarray(ids, np.int32).reshape(-1, 3)
    edge = 8.0 * float(res)
    keep, offs, verts, cols = [], [0], [], []
    for cid in ids:
        zlo = cid[2] * edge
        if not (zlo <= z0 < zlo + edge):
            continue
        h = (int(cid[0]) * 73856093) ^ (int(cid[1]) * 19349663) ^ (int(cid[2]) * 83492791) ^ seed
        rng = np.random.Generator(np.random.PCG64(h & 0x7FFFFFFF))
        n = 12 + int(rng.integers(0, 28))
        p = np.empty((n, 3), np.float64)
        p[:, 0] = cid[0] * edge + rng.random(n) * edge
        p[:, 1] = cid[1] * edge + rng.random(n) * edge
        p[:, 2] = z0
        keep.append(cid)
        verts.append(p.astype(np.float32))
        cols.append(rng.random((n, 3)).astype(np.float32))
        offs.append(offs[-1] + n)
    if not keep:
        return (np.zeros((0, 3), np.int32), np.zeros(1, np.int64), np.zeros((0, 3), np.float32),
                np.zeros((0, 3), np.float32))
    return (np.asarray(keep, np.int32), np.asarray(offs, np.int64), np.concatenate(verts),
            np.concatenate(cols))


def pose_inverse16(pose: np.ndarray) -> np.ndarray:
    """f32(SE3d.inverse().matrix()) of a camera-to-world [R|t] (Patch.cpp:51): double inverse, then cast."""
    p = np.asarray(pose, np.float64).reshape(3, 4)
    R, t = p[:, :3], p[:, 3]
    T = np.eye(4)
    T[:3, :3] = R.T
    T[:3, 3] = -R.T @ t
    return T.astype(np.float32).reshape(16)


def mesh_from_depth(depth, rgba, pose, cam: Camera, res, stride: int = 4, max_chunks: int | None = None):
    """Synthetic per-chunk "meshes" for the atlas stage of an arbitrary frame: every stride-th valid
    pixel is back-projected onto the observed surface and the points are grouped by the chunk they
    fall into (the reference takes marching-cubes vertices, Structure/ChunkManager.cpp:595-1002).
    Vertex colours are the pixel colours / 255.  Chunks are returned in first-seen (row-major) order.
    Returns (chunk ids [m,3] i32, vertex offsets [m+1] i64, verts [nv,3] f32, colors [nv,3] f32)."""
    d = np.asarray(depth, np.float32)[::stride, ::stride].astype(np.float64)
    v, u = np.meshgrid(np.arange(0, cam.height, stride), np.arange(0, cam.width, stride), indexing="ij")
    ok = d > 0
    x = (u[ok] - cam.cx) / cam.fx * d[ok]
    y = (v[ok] - cam.cy) / cam.fy * d[ok]
    pc = np.stack([x, y, d[ok]], 1)
    P = np.asarray(pose, np.float64).reshape(3, 4)
    pw = (pc @ P[:, :3].T + P[:, 3]).astype(np.float32)
    col = np.asarray(rgba)[::stride, ::stride][ok][:, :3].astype(np.float32) / np.float32(255.0)
    edge = np.float32(8.0) * np.float32(res)
    cid = np.floor(pw / edge).astype(np.int32)
    key = (cid[:, 0].astype(np.int64) << 42) ^ (cid[:, 1].astype(np.int64) << 21) ^ cid[:, 2].astype(np.int64)
    _, first, inv = np.unique(key, return_index=True, return_inverse=True)
    order_of_group = np.argsort(np.argsort(first))  # groups numbered by first appearance
    g = order_of_group[inv]
    srt = np.argsort(g, kind="stable")
    counts = np.bincount(g)
    if max_chunks is not None and len(counts) > max_chunks:
        keep = g[srt] < max_chunks
        srt = srt[keep]
        counts = counts[:max_chunks]
    ids = cid[first[np.argsort(first)]][: len(counts)]
    voff = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return ids.astype(np.int32), voff, np.ascontiguousarray(pw[srt]), np.ascontiguousarray(col[srt])
