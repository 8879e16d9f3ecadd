#!/bin/bash
# round 5 (second session): bench.py with the call side taking G steps' sub-clusters per batch (--call-group), G = 1 is rounds 2-5
cd "$(dirname "$0")/.."
for cfg in "--call-group 1" "--call-group 2" "--call-group 3" "--call-group 2 --call-threads 2" "--call-group 3 --call-threads 2" "--call-group 5 --call-threads 2" "--call-group 1"; do
  echo "== $cfg"
  python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline $cfg 2>/tmp/bench_err.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); c=d['config']['call_dp']
print('free HBM before the steps %.0f GB;' % (d['config']['hbm_free_before_the_steps']/1e9), 'value %.0f reads/s, %.1f ms per step | search kernel %.1f ms in the step, %.1f alone | per step: POA %.1f realign %.1f call wall %.1f | batches %s | group alone %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_on_idle_gpu'], c['poa_kernel_ms'], c['realign_kernel_ms'], c['call_wall_ms_per_step'], c['call_batches_in_the_timed_region'], c['group_call_on_idle_gpu']))" || tail -3 /tmp/bench_err.txt
done
