#!/bin/bash
# round 6: does an idle process with a GPU context (as bench.py is while its e2e legs run) slow the chain's binaries down?
set -u
TAG=${TAG:-r06bd}
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/$TAG
python - <<'PY' &
import time, torch
torch.cuda.set_device(0)
x = torch.zeros(1 << 28, dtype=torch.uint8, device="cuda")
streams = [torch.cuda.Stream() for _ in range(24)]
for s in streams:
    with torch.cuda.stream(s):
        x[:1024].add_(1)
torch.cuda.synchronize()
print("idle context up", flush=True)
time.sleep(600)
PY
HOLDER=$!
sleep 20
TAG=$TAG bash tools/r06_chain_like_bench.sh
kill $HOLDER 2>/dev/null
