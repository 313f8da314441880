#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]   -- kernel trace + PMC passes (each in a run of its own) of bench.py on the GPU box
# (--no-secondary: the 64x128 / peaked measurements in front of the headline run the same kernel instantiation and would
#  be averaged into its per-dispatch figures)
set -e
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary "$@" > $out/trace.log 2>&1 || true
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/pmc1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out/pmc1.log 2>&1 || true
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out/pmc2.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc3 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out/pmc3.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc4 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out/pmc4.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES --output-format csv -d $out/pmc5 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out/pmc5.log 2>&1 || true
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $out/pmc6 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out/pmc6.log 2>&1 || true
python3 tools/prof_summary.py $out > $out/summary.txt 2>&1 || true
cat $out/summary.txt
