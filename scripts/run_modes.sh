# does k_jtj_fdp's time depend on where the buffers land? six processes: addresses (host profile line) and kernel times
for i in 1 2 3 4 5 6; do python bench.py --steps 24 --no-cpu-baseline --survey-steps 0 --variant 256 2> gpurun_out/modes_$i.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'fdp', round(d['roofline']['avg_launch_ms'],4), 'lr', round(d['broyden_kernel']['avg_launch_ms'],4), 'gemm', round(d['residual_gemm']['avg_call_ms'],4), 'trial', round(d['trial_residual']['avg_call_ms'],4))"; grep -m1 "FD panel" gpurun_out/modes_$i.err; done
