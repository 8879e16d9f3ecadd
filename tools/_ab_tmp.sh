cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5
B="--steps 20 --warmup 5 --no-cpu-baseline --no-e2e"
show() { python -c "
import json,sys;d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);c=d['config']['call_dp'];print(sys.argv[1].split('/')[-1],round(d['value']),round(d['ms_per_step'],1),'search',round(d['config']['search_ms_per_step'],1),'call_wall',round(c['call_wall_ms_per_step'],1),'poa',c['poa_kernel_ms'],'aln',c['realign_kernel_ms'])" $1 || tail -5 $2; }
for t in 3 2 4 3 2; do
  timeout 600 python bench.py $B --call-threads $t > gpurun_out/s5/b_ct$t.json 2> gpurun_out/s5/b_ct$t.err; show gpurun_out/s5/b_ct$t.json gpurun_out/s5/b_ct$t.err
done
