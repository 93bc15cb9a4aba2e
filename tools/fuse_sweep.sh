#!/bin/bash
# where the step's forcings ride (SZ_FUSE_FORCING: 0 own launch, 1 neighbour launch, 2 narrow launch) vs field size
for n in ${SIZES:-2000 5000 10000 20000 40000 65000}; do
  for m in 0 1 2; do
    SZ_FUSE_FORCING=$m python bench.py --floes $n --steps 60 --warmup 10 --repeats 5 --no-cpu-baseline ${EXTRA} 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('n=$n mode=$m ms/step %.4f  (min %.4f max %.4f)  Mfs/s %.2f' % (j['ms_per_step'], j['ms_per_step_min'], j['ms_per_step_max'], j['value']/1e6))"
  done
done
