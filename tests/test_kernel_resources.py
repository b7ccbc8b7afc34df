"""Register / private-memory budget of the hot kernels, checked at build time (hipcc cross-compiles without a GPU):
a kernel that silently starts to spill, or whose kernel-argument struct ends up in private memory (as the depth-only
instance of k_frame once did: 1.2 KB per lane, 9x slower), still passes every parity test -- this catches it."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "texturefusion_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# kernel name fragment -> (max VGPRs, max scratch bytes per lane); the PRODUCT instances (last template flag false: the
# tuning instances with time stamps and triage cut-offs are not budgeted)
BUDGET = {
    "tf_kernels.hip": {"k_frameILb1ELb0ELb0E": (72, 0), "k_frameILb0ELb0ELb0E": (72, 0), "k_frameILb1ELb1ELb0E": (72, 8),
                       "k_scanENS": (64, 0)},  # (the look-back scan of tf_prepare: eight waves per SIMD)
    # (with the keyframe's colour pass on board: the LDS tables allow four waves per SIMD = 128 VGPRs)
    "tf_group.hip": {"k_integrate_groupILb1ELb0E": (96, 0), "k_integrate_groupILb0ELb0E": (96, 0),
                     "k_integrate_groupILb1ELb1E": (128, 0), "k_integrate_groupILb0ELb1E": (128, 0)},
    # (the filter's two forms -- wave per entry / workgroup batches -- are two kernels; the instances that carry the previous
    # frame's patch stage (the keyframe unit) are compiled for 6 waves per SIMD: 80 VGPRs.  Round 6: no instance that carries
    # the patch stage owns private memory any more -- what they spilled were loop invariants of the patch loop (float / double
    # / reciprocal forms of the slot and atlas sizes, a lane's byte offsets), now kept opaque so that they are recomputed at
    # their uses instead of being reloaded -- a dependent round trip each -- from scratch)
    "tf_mesh.hip": {"k_meshILi128ELb0E": (80, 0), "k_mesh_filterILb1ELb0ELb0E": (72, 0),
                    "k_mesh_filterILb0ELb0ELb0E": (80, 0), "k_mesh_filterILb1ELb1ELb0E": (72, 0),
                    # (the batch form with the patch stage on board -- the keyframe unit on hall-sized lists: a 20-byte stack slot the
                    # compiler reserves and then folds away -- its code holds no scratch instruction)
                    "k_mesh_filterILb0ELb1ELb0E": (80, 24)},
    "tf_atlas.hip": {"k_patchILb1ELb1ELb1E": (80, 0)},  # one patch per wave, two 64-vertex blocks in registers: 6 waves per SIMD
}


def _usage(src):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
           "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", os.devnull]
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and name:
            out[name][m.group(1).split(" ")[0]] = int(m.group(2))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src", sorted(BUDGET))
def test_hot_kernels_stay_within_their_register_budget(src):
    usage = _usage(src)
    for frag, (max_vgpr, max_scratch) in BUDGET[src].items():
        hits = {k: v for k, v in usage.items() if frag in k}
        assert hits, "kernel %s not found in %s" % (frag, src)
        for k, v in hits.items():
            assert v["ScratchSize"] <= max_scratch, "%s uses %d B/lane of private memory" % (k, v["ScratchSize"])
            assert v["VGPRs"] <= max_vgpr, "%s uses %d VGPRs (budget %d)" % (k, v["VGPRs"], max_vgpr)
