#!/bin/bash
# developer sweep: search throughput vs resident blocks and segments per read
for seg in 4 8 16; do for blk in 512 768 1024 1536 2048; do
  v=$(SVDSS_SEGMENTS=$seg SVDSS_BLOCKS=$blk python bench.py --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('reads_redone_unsegmented'))")
  echo "seg=$seg blocks=$blk -> $v"
done; done
