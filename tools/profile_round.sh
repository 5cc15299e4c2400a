#!/bin/bash
# Round profile (run on the GPU box via gpurun): for each named command
#   1. rocprofv3 --kernel-trace --stats          -> per-kernel average durations
#   2. separate --pmc passes (never combined with tracing): HBM-side traffic, L2 hit rate, SQ issue / wait, LDS conflicts
# and one summary JSON per command under gpurun_out/profiles_<tag>/ (copy what should be judged into profiles/).
# usage: tools/profile_round.sh <tag> [case ...]     cases: bench bench20 linear512 sweep1024 linear1024 general512 axis2_512 prefilter512 prefilter1024 (default: all but bench20)
tag=${1:-r02}; shift
cases=${@:-bench linear512 sweep1024 linear1024 general512 axis2_512 prefilter512 prefilter1024}
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/profiles_$tag
mkdir -p $out
passes=(
 "FETCH_SIZE GRBM_GUI_ACTIVE"
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
)
profile() {   # name, then the python command line (the program itself follows `--`: no shell, no env wrapper)
  name=$1; shift
  d=$out/$name; rm -rf $d; mkdir -p $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -- python3 "$@" > $d/stdout_under_trace.txt 2> $d/trace.log
  i=0
  for pass in "${passes[@]}"; do
    rocprofv3 --pmc $pass --output-format csv -d $d/pmc$i -- python3 "$@" > /dev/null 2> $d/pmc$i.log
    i=$((i+1))
  done
  python3 tools/profile_summary.py $d $tag $name > $d/summary_head.txt
  cp $d/${tag}_${name}_summary.json $out/ 2>/dev/null
  cp $d/trace/*/*kernel_stats.csv $out/${tag}_${name}_kernel_stats.csv 2>/dev/null
  cp $d/stdout_under_trace.txt $out/${tag}_${name}_stdout_under_trace.txt
  echo "== $name"; head -c 1500 $d/summary_head.txt
}
for c in $cases; do
  case $c in
    # (--no-extra-1024: the 1024^3 trilinear launches of `extra` would share a kernel-stats row with the 512^3 ones)
    bench)         profile bench bench.py --steps 180 --warmup 5 --no-cpu-baseline --no-extra-1024 ;;
    linear512)     profile linear512 tools/prof_case.py --size 512 --interp linear --sweep 1 --iters 180 ;;     # BASELINE config #2
    sweep1024)     profile sweep1024 bench.py --size 1024 --steps 60 --warmup 3 --no-cpu-baseline ;;
    linear1024)    profile linear1024 tools/prof_case.py --size 1024 --interp linear --sweep 6 --iters 30 ;;    # the north star's 1024^3 trilinear sweep
    general512)    profile general512_linear tools/prof_case.py --size 512 --interp linear --general --iters 30
                   profile general512_cubic tools/prof_case.py --size 512 --interp filt_bspline --general --iters 30 ;;
    # the driver's exact command: kernel stats and `ms_per_step` of its BENCH record are directly comparable
    bench20)       profile bench20 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-1024 ;;
    # rotations about array axis 2 (the row kernel, kind 10), 33 degrees
    axis2_512)     profile axis2_512_linear tools/prof_case.py --size 512 --interp linear --axis2 --angle 33 --iters 30
                   profile axis2_512_cubic tools/prof_case.py --size 512 --interp filt_bspline --axis2 --angle 33 --iters 30 ;;
    prefilter512)  profile prefilter512 tools/prefilter_time.py 512 ;;
    prefilter1024) profile prefilter1024 tools/prefilter_time.py 1024 ;;
  esac
done
ls $out
