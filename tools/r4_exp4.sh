#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r4_exp4; rm -rf $out; mkdir -p $out
python tools/exp_inflate_host.py > $out/inflate_host.txt 2>&1
for cfg in "4096 2048" "32768 16384"; do
  tag=$(echo $cfg | tr ' ' '_')
  python tools/exp_lane_small.py $cfg > $out/lane_$tag.txt 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $out/p_$tag/pmc_sq -o t -- python tools/exp_lane_small.py $cfg > /dev/null 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU -d $out/p_$tag/pmc_lds -o t -- python tools/exp_lane_small.py $cfg > /dev/null 2>&1
  python tools/summarize_prof.py $out/p_$tag > $out/pmc_$tag.txt 2>&1
  find $out/p_$tag -name "*.csv" -delete
done
# clock while k_compress runs: poll the SMI during a long headline loop
( python bench.py --no-secondary --no-archive --no-end-to-end --cpu-seconds 0 --verify 0 --steps 1500 --warmup 5 > $out/long_bench.json 2>/dev/null & )
sleep 14
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2; sleep 0.7; done > $out/sclk_during_compress.txt 2>&1
wait
sleep 6
cat $out/lane_*.txt; tail -4 $out/inflate_host.txt
