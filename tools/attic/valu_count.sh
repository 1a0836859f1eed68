#!/bin/bash
# dynamic VALU wave instructions per launch of a kernel:  tools/attic/valu_count.sh <kernel substring> <bench args...>
k=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_v
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_v -o run -- python3 bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-also > gpurun_out/pmc_v.log 2>&1
python3 - "$k" <<EOF
import csv,glob,collections,sys
acc=0.0; n=set()
for f in glob.glob("gpurun_out/pmc_v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            acc+=float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
print("SQ_INSTS_VALU per launch", acc/max(1,len(n)))
EOF
