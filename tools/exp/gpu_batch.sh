#!/bin/bash
# One parametrised GPU batch script (replaces round 5's tools/exp/r05_batch*.sh): `gpu_batch.sh <section> [<section> ...]`, every
# section writes gpurun_out/r06_<section>.txt.  Sections run under their own `timeout` so that a hung kernel cannot hold the box.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for sec in "$@"; do
  out=gpurun_out/r06_$sec.txt
  case $sec in
    w4_tests)   # correctness of the one-wave-per-SIMD NT GEMM (poisoned LDS, bit-identity with the 8-wave kernel)
      timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -k "pre_split_a or pre_packed_one_term" -s 2>&1 | tail -40 > $out ;;
    w4_bench)   # both kernels, two-term and packed, at the three large resolutions
      timeout 600 python tools/exp/gemm_nt_bench.py --entries gemm_f16x2_pre,gemm_f16x2_pre_w4,gemm_f16p_pre,gemm_f16p_pre_w4 > $out 2>&1 ;;
    gemm_abl)   # part-removal ablation of today's 8-wave pre-split kernel + cycle stamps, and of the w4 kernel
      { for m in 1 2 8 16 14; do
          DSEE_LIB=tools/exp/libabl_$m.so timeout 300 python tools/exp/gemm_nt_bench.py --entries gemm_f16x2_pre --shapes 256 --tag "w8 mask $m"
        done
        DSEE_LIB=tools/exp/libabl_32.so timeout 300 python tools/exp/gemm_nt_bench.py --entries gemm_f16x2_pre --shapes 256 --stamps --tag "w8 stamps"
        for m in a1 a2 a8 a16 a10; do
          DSEE_LIB=tools/exp/libw4_$m.so timeout 300 python tools/exp/gemm_nt_bench.py --entries gemm_f16x2_pre_w4 --shapes 256 --tag "w4 $m"
        done; } > $out 2>&1 ;;
    w4_var)     # w4 measurement builds named in W4_VARIANTS (tools/exp/libw4_<name>.so), 256^2 shape
      { for m in $W4_VARIANTS; do
          DSEE_LIB=tools/exp/libw4_$m.so timeout 300 python tools/exp/gemm_nt_bench.py --entries gemm_f16x2_pre_w4 --shapes 256 --tag "w4 $m" 2>&1 | grep -v amdgpu.ids
        done; } > $out 2>&1 ;;
    w4_stamps)  # cycle stamps of the w4 kernel (builds named in W4_VARIANTS, compiled with DSEE_W4_ABL & 32)
      { for m in $W4_VARIANTS; do
          DSEE_LIB=tools/exp/libw4_$m.so timeout 300 python tools/exp/gemm_nt_bench.py --entries gemm_f16x2_pre_w4 --shapes 256 --stamps --tag "w4 $m" 2>&1 | grep -v amdgpu.ids
        done; } > $out 2>&1 ;;
    power)      # socket power / shader clock under the NT GEMM kernels (POWER_RUNS: "lib:entry[:zero]" items; lib '-' = shipped)
      { rocm-smi -P -c 2>&1 | head -12
        for r in $POWER_RUNS; do
          IFS=: read lib entry zero real <<< "$r"
          [ "$lib" = "-" ] && unset DSEE_LIB || export DSEE_LIB=tools/exp/$lib.so
          TAG=$lib ZERO=${zero:-0} REAL=${real:-0} timeout 120 python tools/exp/power_probe.py $entry 3 2>&1 | grep -v amdgpu.ids
        done; unset DSEE_LIB; } > $out 2>&1 ;;
    bench_power) # the driver's command with rocm-smi sampled beside it (socket power, shader clock)
      { ( while true; do rocm-smi -P -c 2>/dev/null | grep -E "Power \(W\)|sclk" | tr '\n' ' '; echo; sleep 0.2; done ) > gpurun_out/r06_bench_power_samples.txt &
        SMI=$!
        timeout 900 python bench.py --steps 40 --warmup 5 --no-f32-run --no-cpu-baseline ${BENCH_ARGS:-} 2>gpurun_out/r06_bench_power.err
        kill $SMI; } > $out 2>&1
      cp $out gpurun_out/r06_bench_power_${BENCH_TAG:-default}.txt
      python - <<'PY' >> gpurun_out/r06_bench_power_${BENCH_TAG:-default}.txt
import re
rows=[l for l in open("gpurun_out/r06_bench_power_samples.txt") if "Power" in l]
pw=[float(m.group(1)) for l in rows for m in [re.search(r"Power \(W\):\s*([\d.]+)",l)] if m]
ck=[int(m.group(1)) for l in rows for m in [re.search(r"sclk clock level:.*?\((\d+)Mhz\)",l)] if m]
hot=[(p,c) for p,c in zip(pw,ck) if p>600]
print("rocm-smi samples: %d, of which %d above 600 W: power mean %.0f max %.0f W, sclk mean %.0f min %d max %d MHz" % (len(pw), len(hot), sum(p for p,_ in hot)/max(1,len(hot)), max(pw or [0]), sum(c for _,c in hot)/max(1,len(hot)), min([c for _,c in hot] or [0]), max([c for _,c in hot] or [0])))
PY
      python -c "
import json,sys
for l in open('gpurun_out/r06_bench_power_${BENCH_TAG:-default}.txt'):
    if l.startswith('{'): d=json.loads(l); print('${BENCH_TAG:-default}', round(d['value'],2), 'img/s', round(d['ms_per_step'],2), 'ms')
    elif l.startswith('rocm-smi'): print(l.strip())
" > $out ;;
    kstats)     # rocprofv3 kernel statistics of the step (5 steps + 2 warm-up + the eager / capture steps), BENCH_TAG / BENCH_ARGS
      ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$$ && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$$ -o ev -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-run ${BENCH_ARGS:-} > /dev/null 2>&1
        python $OLDPWD/tools/rocpd_summary.py /tmp/prof_$$/ev_results.db 400 ) > gpurun_out/r06_kstats_${BENCH_TAG:-default}.md 2>&1
      head -45 gpurun_out/r06_kstats_${BENCH_TAG:-default}.md > $out; tail -2 gpurun_out/r06_kstats_${BENCH_TAG:-default}.md >> $out ;;
    kclock)     # shader clock per kernel inside the step (one PMC pass: GRBM_GUI_ACTIVE / duration)
      ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmcclk_$$ && timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d /tmp/pmcclk_$$ -o p --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-run --no-graphs ${BENCH_ARGS:-} > /dev/null 2>&1
        python $OLDPWD/tools/pmc_clock.py /tmp/pmcclk_$$/*counter_collection.csv ) > $out 2>&1 ;;
    step_shapes) # per-geometry time of the direct convolutions + per-autograd-function time of one eager step
      { timeout 600 python tools/step_shapes.py 2>&1 | grep -v "amdgpu.ids\|Warning\|self.sr_model\|^lr G"
        timeout 600 python tools/step_functions.py 60 2>&1 | grep -v "amdgpu.ids\|Warning\|self.sr_model\|^lr G"; } > $out 2>&1 ;;
    alternate)  # the GEMM back to back vs alternating with light kernels (power-state transitions)
      { timeout 300 python tools/exp/gemm_alternate.py gemm_f16x2_pre_w4; timeout 300 python tools/exp/gemm_alternate.py gemm_f16x2_pre; } 2>&1 | grep -v amdgpu.ids > $out ;;
    pytest)     # PYTEST_ARGS: a selection of the GPU suite
      DSEE_TEST_DURATIONS=gpurun_out/r06_durations.txt timeout ${PYTEST_TIMEOUT:-1500} python -m pytest $PYTEST_ARGS 2>&1 | grep -v amdgpu.ids | tail -${PYTEST_TAIL:-60} > $out ;;
    mfma_power) # dense 16-bit MFMA rate the socket power cap allows (operands in registers, no memory traffic), rocm-smi beside it
      { for a in "fp16 random" "bf16 random" "bf16 zeros"; do
          ( sleep 2.5; rocm-smi -P -c 2>/dev/null | grep -E "Power \(W\)|sclk" | tr '\n' ' '; echo ) &
          timeout 60 tools/exp/mfma_power_probe $a 4; wait
        done; } > $out 2>&1 ;;
    fused_bench) # the fused SPADE / SEAN forward alone: 8-wave kernel vs the one-wave-per-SIMD kernel (FUSED_LIBS: measurement builds)
      { timeout 300 python tools/exp/fused_kernel_bench.py
        FUSED_ENTRY=spade_fused_fwd_w4 timeout 300 python tools/exp/fused_kernel_bench.py
        for l in ${FUSED_LIBS:-}; do DSEE_LIB=tools/exp/$l.so FUSED_ENTRY=spade_fused_fwd_w4 timeout 300 python tools/exp/fused_kernel_bench.py; done; } 2>&1 | grep -v amdgpu.ids > $out ;;
    bench)      # the driver's command
      timeout 900 python bench.py --steps 20 --warmup 5 > $out 2>gpurun_out/r06_bench.err ;;
    bench_quick)
      timeout 900 python bench.py --steps 20 --warmup 5 --no-f32-run --no-cpu-baseline > $out 2>gpurun_out/r06_bench_quick.err ;;
    *) echo "unknown section $sec" ;;
  esac
  echo "== $sec"; tail -30 $out
done
