"""Offline RGB-D sequences in the layout the reference's DatasetWrapper reads (Tools/DatasetWrapper.hpp:55-263,
InputMode = 0) -- BASELINE.json configs[0], plumbing only.

    <dir>/associate.txt    lines "tRGB rgb_path tDepth depth_path"
    <dir>/calib.txt        one line "width height fx fy cx cy d0 d1 d2 d3 d4 depth_scale maximum_depth"
    <dir>/rgb/*.png        8-bit RGB,  <dir>/depth/*.png  16-bit depth (value / depth_scale = metres)
    <dir>/groundtruth.txt  "t tx ty tz qx qy qz qw" per frame (TUM convention).  The reference gets its poses from
                           tracking and leaves this reader commented out (:104-133); the plumbing here uses it as the
                           pose source because the SLAM front end is out of scope.

load_frame() restates DatasetWrapper::LoadSingleFrame + framePreprocess (:164-263): depth above maximum_depth *
depth_scale -> 0, refined_depth = depth / depth_scale as f32 metres, weight = 0.  The cv::bilateralFilter step
(:231-233) is OpenCV arithmetic and is bypassed (out of scope).  PNG files are written / read with a
minimal zlib codec of this module (8-bit RGB and 16-bit grey, no interlace), so nothing outside the standard
library is needed; the real example archives of the reference's README are not reachable offline."""
from __future__ import annotations

import math
import os
import struct
import zlib

import numpy as np

_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_png(path: str, img: np.ndarray) -> None:
    """img: u8 [H, W, 3] (RGB) or u16 [H, W] (grey)."""
    img = np.ascontiguousarray(img)
    if img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3:
        depth, ctype, rows = 8, 2, img.reshape(img.shape[0], -1)
    elif img.dtype == np.uint16 and img.ndim == 2:
        depth, ctype, rows = 16, 0, img.astype(">u2").view(np.uint8).reshape(img.shape[0], -1)
    else:
        raise ValueError("write_png: u8 RGB or u16 grey only")
    h, w = img.shape[0], img.shape[1]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rows], axis=1).tobytes()  # filter type 0 on every row
    with open(path, "wb") as fh:
        fh.write(_PNG_SIG + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                 _chunk(b"IDAT", zlib.compress(raw, 3)) + _chunk(b"IEND", b""))


def read_png(path: str) -> np.ndarray:
    """-> u8 [H, W, 3] or u16 [H, W]; 8/16-bit grey, RGB and RGBA (alpha dropped), no interlace, all five filters."""
    with open(path, "rb") as fh:
        data = fh.read()
    if data[:8] != _PNG_SIG:
        raise ValueError("%s: not a PNG file" % path)
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    if interlace or depth not in (8, 16) or ctype not in (0, 2, 6):
        raise ValueError("%s: unsupported PNG flavour (depth %d, colour type %d, interlace %d)" % (path, depth, ctype, interlace))
    nch = {0: 1, 2: 3, 6: 4}[ctype]
    bpp = nch * depth // 8
    stride = w * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        elif f == 1:  # Sub: a running sum per byte lane of the pixel
            cur = line.reshape(-1, bpp).cumsum(axis=0).reshape(-1) & 255
        else:         # Average / Paeth: sequential in x
            cur = np.zeros(stride, np.int32)
            for x in range(stride):
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                if f == 3:
                    p = (a + b) >> 1
                else:
                    c = prev[x - bpp] if x >= bpp else 0
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + p) & 255
        out[y] = cur
        prev = cur
    if depth == 16:
        img = out.view(">u2").astype(np.uint16).reshape(h, w, nch)
    else:
        img = out.reshape(h, w, nch)
    if nch == 1:
        return img[..., 0]
    return np.ascontiguousarray(img[..., :3])


def _quat_from_R(R):
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        return ((R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s)
    i = int(np.argmax([R[0, 0], R[1, 1], R[2, 2]]))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
    q = [0.0, 0.0, 0.0, 0.0]
    q[i] = 0.25 * s
    q[j] = (R[j, i] + R[i, j]) / s
    q[k] = (R[k, i] + R[i, k]) / s
    q[3] = (R[k, j] - R[j, k]) / s
    return tuple(q)


def _R_from_quat(qx, qy, qz, qw):
    n = math.sqrt(qx * qx + qy * qy + qz * qz + qw * qw)
    qx, qy, qz, qw = qx / n, qy / n, qz / n, qw / n
    return np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                     [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                     [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]], np.float64)


def write_sequence(folder: str, frames, cam, depth_scale: float = 5000.0, maximum_depth: float = 8.0, fps: float = 30.0):
    """frames: iterable of (depth f32 [H, W] metres, rgba or rgb u8, _, pose f32 [3, 4]) -- what synth.room_frame returns."""
    os.makedirs(os.path.join(folder, "rgb"), exist_ok=True)
    os.makedirs(os.path.join(folder, "depth"), exist_ok=True)
    with open(os.path.join(folder, "calib.txt"), "w") as fh:
        fh.write("%d %d %.9g %.9g %.9g %.9g 0 0 0 0 0 %.9g %.9g\n"
                 % (cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, depth_scale, maximum_depth))
    assoc = open(os.path.join(folder, "associate.txt"), "w")
    gt = open(os.path.join(folder, "groundtruth.txt"), "w")
    gt.write("# timestamp tx ty tz qx qy qz qw\n")
    for k, f in enumerate(frames):
        depth, col, pose = f[0], f[1], f[3]
        t = k / fps
        name = "%06d.png" % k
        write_png(os.path.join(folder, "rgb", name), np.ascontiguousarray(col[..., :3]))
        d16 = np.clip(np.rint(depth.astype(np.float64) * depth_scale), 0, 65535).astype(np.uint16)
        write_png(os.path.join(folder, "depth", name), d16)
        assoc.write("%.6f rgb/%s %.6f depth/%s\n" % (t, name, t, name))
        q = _quat_from_R(pose[:, :3].astype(np.float64))
        gt.write("%.6f %.9g %.9g %.9g %.12g %.12g %.12g %.12g\n" % ((t,) + tuple(float(x) for x in pose[:, 3]) + q))
    assoc.close()
    gt.close()


class Sequence:
    """DatasetWrapper::init (:55-162): associate.txt + calib.txt (+ groundtruth.txt for the poses)."""

    def __init__(self, folder: str):
        self.folder = folder
        self.rgb_files, self.depth_files, self.time_stamp = [], [], []
        with open(os.path.join(folder, "associate.txt")) as fh:
            for line in fh:
                tok = line.split()
                if len(tok) == 4:  # lines with another token count are skipped, as in the reference
                    self.rgb_files.append(os.path.join(folder, tok[1]))
                    self.depth_files.append(os.path.join(folder, tok[3]))
                    self.time_stamp.append((float(tok[0]) + float(tok[2])) / 2)
        with open(os.path.join(folder, "calib.txt")) as fh:
            tok = fh.readline().split()
        if len(tok) != 13:
            raise ValueError("calib.txt: 13 numbers expected (error in loading parameters)")
        v = [float(x) for x in tok]
        self.width, self.height = int(v[0]), int(v[1])
        self.fx, self.fy, self.cx, self.cy = v[2:6]
        self.distortion = v[6:11]
        self.depth_scale, self.maximum_depth = v[11], v[12]
        self.poses = []
        gt = os.path.join(folder, "groundtruth.txt")
        if os.path.exists(gt):
            with open(gt) as fh:
                for line in fh:
                    tok = line.split()
                    if len(tok) == 8 and not line.startswith("#"):
                        x = [float(s) for s in tok]
                        P = np.concatenate([_R_from_quat(*x[4:8]), np.array(x[1:4]).reshape(3, 1)], axis=1)
                        self.poses.append(P.astype(np.float32))

    def __len__(self):
        return len(self.rgb_files)

    def camera(self, near: float = 0.01, far: float = 5.0):
        from . import synth
        return synth.Camera(width=self.width, height=self.height, fx=self.fx, fy=self.fy, cx=self.cx, cy=self.cy,
                            near=near, far=far)

    def load_frame(self, i: int):
        """LoadSingleFrame + framePreprocess without the bilateral filter -> (refined_depth f32 m, rgba u8 with
        A = 1, weight f32 zeros, pose or None)."""
        rgb = read_png(self.rgb_files[i])
        d16 = read_png(self.depth_files[i])
        if rgb.ndim != 3 or d16.ndim != 2:
            raise ValueError("load image error: %s %s" % (self.rgb_files[i], self.depth_files[i]))
        d16 = d16.copy()
        d16[d16.astype(np.float64) > self.maximum_depth * self.depth_scale] = 0
        refined = (d16.astype(np.float32) / np.float32(self.depth_scale)).astype(np.float32)
        rgba = np.concatenate([rgb, np.ones(rgb.shape[:2] + (1,), np.uint8)], axis=2)
        pose = self.poses[i] if i < len(self.poses) else None
        return refined, np.ascontiguousarray(rgba), np.zeros_like(refined), pose
