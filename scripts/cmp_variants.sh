#!/bin/bash
# GPU box: the headline step of bench.py with the product library and with experiment builds of it, one after the other on the
# same box.  usage: cmp_variants.sh <tag> <variant>...   ("base" names the product library; BENCH_ARGS adds bench.py arguments; results in gpurun_out/<tag>_<variant>.json)
tag=$1; shift
for v in "$@"; do
    if [ "$v" = base ] || [ "${v%%[0-9]*}" = base ]; then unset PTCORE_LIB; else export PTCORE_LIB=$PWD/pathtracer_amd/libptcore_exp_$v.so; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps 8 $BENCH_ARGS > gpurun_out/${tag}_$v.json 2> gpurun_out/${tag}_$v.err || { echo "$v failed"; tail -3 gpurun_out/${tag}_$v.err; exit 1; }
    python - "$v" gpurun_out/${tag}_$v.json <<'EOF'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"{sys.argv[1]:>10}  {d['value']:8.1f} Mrays/s  step {d['ms_per_step']:.3f} ms  kernel {d['roofline']['kernel_ms']:.3f} ms")
EOF
done
