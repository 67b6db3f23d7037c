#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c15; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_llama_decode.py -x -q -s -k "graph_reuse" > $O/test_reuse.log 2>&1; echo "rc=$?" >> $O/test_reuse.log )
grep -E "passed|failed|FAILED|Error|rc=|assert" $O/test_reuse.log | head -20
python - > $O/plugin.log 2>&1 <<'PY'
import sys, os, json
sys.path.insert(0, os.getcwd())
import bench
import prima_cpp_amd.engine as E
pd, path = bench.plugin_decode("/tmp")
print(json.dumps(pd))
if path and os.path.exists(path): os.unlink(path)
print(json.dumps(bench.plugin_decode_70b("/tmp", E.LLAMA3_70B, None)))
PY
grep -o '"tokens_per_s": [0-9.]*\|"graph_reuse_patch": {[^}]*}\|"tokens_per_s_80_layers": [0-9.]*\|"ms_per_layer": [0-9.]*\|"ms_fixed": [0-9.]*' $O/plugin.log | head -30
tail -3 $O/plugin.log | cut -c1-300
