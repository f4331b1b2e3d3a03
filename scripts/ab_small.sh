#!/bin/bash
# rounds enqueued ahead (default) against MIR_LSQ_VARIANT_NO_PIPELINE = 4194304 at the per-rank sizes of a strong-scaled cfg 3 (run on the GPU box)
for M in 125000 250000; do
  for v in 0 4194304; do
    BENCH_M=$M python bench.py --steps 200 --no-cpu-baseline --survey-steps 0 --no-kernel-timing --variant $v 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1])
print('rows $M variant=$v'.ljust(30), 'value %9.1f it/s   %7.3f ms per solve' % (d['value'], d['ms_per_step']))"
  done
done
