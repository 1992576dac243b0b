#!/bin/bash
# GPU box: FETCH_SIZE of three access patterns with a known byte count (4.29 GB each).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/calib; mkdir -p $OUT
[ -x scripts/calib/calib_fetch ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o scripts/calib/calib_fetch scripts/calib/calib_fetch.hip || exit 1
for P in 0 1 2; do
  timeout -k 10 60 scripts/calib/calib_fetch $P > $OUT/time_$P.txt 2>&1 || exit 1
  timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$P -o c -- scripts/calib/calib_fetch $P > $OUT/pmc_$P.txt 2>&1 || exit 1
  timeout -s KILL 60 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmcb_$P -o c -- scripts/calib/calib_fetch $P > $OUT/pmcb_$P.txt 2>&1 || echo "pmcb $P failed"
done
cat $OUT/time_*.txt
grep -h k_gather $OUT/pmc_*/c_counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}'
grep -h k_gather $OUT/pmcb_*/c_counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}'
