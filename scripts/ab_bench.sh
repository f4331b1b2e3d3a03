#!/bin/bash
# A/B of the host-side mechanisms on the GPU box: pipelined rounds on/off, kernel-timing events on/off, cfg3 and cfg2.
# usage: bash scripts/ab_bench.sh   (prints value / ms per solve per case)
for cfg in cfg3 cfg2; do
  for v in 0 4194304; do   # 4194304 = MIR_LSQ_VARIANT_NO_PIPELINE (rounds enqueued ahead are the default for J <= 32 MB)
    for t in "" "--no-kernel-timing"; do
      python bench.py --config $cfg --steps 60 --no-cpu-baseline --survey-steps 0 --variant $v $t 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1])
print('$cfg variant=$v $t'.ljust(44), 'value %9.1f it/s   %7.3f ms per solve' % (d['value'], d['ms_per_step']))"
    done
  done
done
