cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06al
for args in "--call-threads 1" "--call-threads 2" "--call-threads 3" "--call-threads 4" "--call-threads 3 --search-threads 2" "--call-threads 2 --search-threads 2" "--no-call-dp"; do
  python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-e2e $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['config'].get('call_dp') or {}
print('$args: value %.0f step %.1f search-in-step %.1f alone %.1f call wall/step %s poa %s aln %s' % (d['value'], d['ms_per_step'], d['config']['search_ms_per_step'], d['config']['search_kernel_ms_on_idle_gpu'], c.get('call_wall_ms_per_step'), c.get('poa_kernel_ms'), c.get('realign_kernel_ms')))
" >> gpurun_out/r06al/threads.txt
done
cat gpurun_out/r06al/threads.txt
