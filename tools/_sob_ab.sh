#!/bin/bash
mkdir -p gpurun_out/r04V
python -m pytest tests -x -q -m gpu -k "sobolev or Sobolev" > gpurun_out/r04V/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r04V/tests.log
for rep in 1 2 3; do
  python bench.py --no-secondary --workload sobolev 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])" >> gpurun_out/r04V/ab.txt
done
python bench.py --no-secondary --workload sobolev --size 512 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('512', d['value'], d['ms_per_step'], d['roofline']['frac'])" >> gpurun_out/r04V/ab.txt
tail -n 3 gpurun_out/r04V/tests.log; cat gpurun_out/r04V/ab.txt
