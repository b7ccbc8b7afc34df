#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_15; mkdir -p $O
TF_MESH_DBG=9 timeout 300 python tools/stamps.py mesh > $O/stamps_mesh.txt 2>&1; cat $O/stamps_mesh.txt | tail -16
