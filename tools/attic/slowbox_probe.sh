#!/bin/bash
# Is a slow box slow for the sampler's STORE PATTERN alone?  headline bench, torch fill, and the
# stores-only chunk walk (tools/store_pattern6.hip: 1 wave per workgroup, tpc tiles, 12 waves / CU) in one call
python bench.py --no-also --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', round(d['ms_per_step'],4), 'ms', round(d['roofline']['frac'],3))"
python tools/stream_ceiling.py 2>/dev/null | grep "fill"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/store_pattern6.hip -o /tmp/sp6 2>/dev/null && /tmp/sp6 | grep "tpc= 8 w/CU=12\|tpc= 4 w/CU=12\|tpc= 8 w/CU=32\|tpc=32 w/CU=12\|64 KB\|1048576 KB"
