#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c19; mkdir -p $O
( timeout 400 python bench.py --section long_context:llama3-70b > $O/long.log 2>&1 ); python - <<'PY'
import json
for ln in open("gpurun_out/c19/long.log"):
    if ln.startswith("{"):
        d = json.loads(ln)
        print(json.dumps(d.get("contexts")))
        print(json.dumps(d.get("attention_kernel", {}).get("cells")))
        print(json.dumps(d.get("attention_kernel_q8_0_kv")))
PY
tail -3 $O/long.log | cut -c1-300
