cd /root/repo
for cfg in "4 256" "6 128" "8 128" "6 192"; do
  set -- $cfg
  echo "## feeders $1, batch $2 MB"
  SVDSS_SEARCH_FEEDERS=$1 SVDSS_BAM_BATCH_MB=$2 python bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-e2e-call 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in ('e2e', 'e2e_smoothed', 'e2e_wg'):
    v = d.get(k, {})
    print(k, v.get('streaming_s_runs'), round(v.get('reads_per_s_streaming', 0)))
"
done
