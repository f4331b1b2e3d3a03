for i in 1 2 3; do python bench.py --steps 60 --no-cpu-baseline --survey-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), 'fdp', round(d['roofline']['avg_launch_ms'],4), 'lr', round(d['broyden_kernel']['avg_launch_ms'],4), 'gemm', round(d['residual_gemm']['avg_call_ms'],4), 'trial', round(d['trial_residual']['avg_call_ms'],4))"; done
