#!/bin/bash
O=gpurun_out/r04R; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-secondary --size 512 > $O/bench_512.json 2>/dev/null
python bench.py --no-secondary --data depth > $O/bench_depth_256.json 2>/dev/null
python bench.py --no-secondary --data depth --size 512 > $O/bench_depth_512.json 2>/dev/null
python bench.py --no-secondary --workload sobolev > $O/bench_sobolev.json 2>/dev/null
python tools/host_timeline.py 256 > $O/host_timeline.txt 2>&1
HALO=8 ITERS=50 FIXED_ONLY=1 python tools/slab_nccl_loopback.py 256 2>&1 | grep -E "iterations|host time|single|compact" > $O/loopback.txt
python -c "
import json
for n in ('default','512','depth_256','depth_512','sobolev'):
    d=json.load(open('$O/bench_%s.json'%n)); print(n, '%.1f G'%(d['value']/1e9), '%.3f ms'%d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'))
d=json.load(open('$O/bench_default.json'))
for s in d.get('secondary',[]): print('  ', s.get('workload'), s.get('ms_per_step'), s.get('roofline',{}).get('frac'), s.get('us_per_iteration'))
"
