#!/bin/bash
# A/B of the two-stream finite-difference refresh on the GPU box: cfg 3 with 0 (one sweep), 2, 4, 8, 16
# row windows. Prints value, ms per solve and the event-timed refresh (caller GEMM + fused kernel; in window mode both are
# inside jtj_fd_kernel). usage: bash scripts/ab_windows.sh > gpurun_out/ab_windows.txt
for w in 0 2 4 8 16; do
  python bench.py --steps 40 --warmup 5 --survey-steps 0 --no-cpu-baseline --no-host-callback --fd-windows $w "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1])
c = d['config']; t = c['time_split_ms_per_solve']; r = c['jacobian_full_per_solve']
print('windows %2d' % $w, ' value %8.1f it/s  %6.3f ms per solve' % (d['value'], d['ms_per_step']),
      '  refresh (caller FD kernels + fused FD kernel) %.3f ms each' % ((t['caller_fd_callbacks'] + t['jtj_fd_kernel']) / r),
      '  min/median/max step', ['%.3f' % v for v in d['ms_per_step_uninstrumented_min_median_max']])"
done
