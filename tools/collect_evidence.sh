#!/bin/bash
# The round's measured evidence in one GPU call:  bash tools/collect_evidence.sh <out dir under gpurun_out> <file prefix>
# (S2E_GIT_HEAD = the commit being measured; there is no .git on the GPU box).  Order matters: the PMC passes come first and
# land in profiles/r06/pmc of the box's copy, so the bench line that follows quotes the traffic of THIS build.
set -u
O=${1:-gpurun_out/evidence}; P=${2:-x}
R=$(pwd); mkdir -p "$R/$O/pmc"
export TMPDIR=/tmp
B="$R/bench.py"
cd /tmp
PMC="--steps 2 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-events --no-extras"
(cd "$R" && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python3 bench.py $PMC > /dev/null 2>&1)
(cd "$R" && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python3 bench.py $PMC > /dev/null 2>&1)
(cd "$R" && python3 tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w profiles/r06/pmc > /dev/null && cp profiles/r06/pmc/hbm_traffic* "$O/pmc/")
(cd "$R" && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
    SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_sq -o sq -- python3 bench.py $PMC > /dev/null 2>&1
 python3 tools/pmc_sq.py /tmp/pmc_sq "$O/pmc/sq_counters_patch_kernels.txt" > /dev/null)
cd "$R"
# the same command under the kernel trace first (its per-family kernel time is quoted by the bench line), then the bench line as the
# driver runs it (cpu_baseline legs included)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras \
    > "$O/${P}_bench_under_rocprof.json" 2>/dev/null
cp /tmp/kt/*/*kernel_stats.csv "$O/${P}_kernel_stats.csv" 2>/dev/null || cp /tmp/kt/*kernel_stats.csv "$O/${P}_kernel_stats.csv"
python3 tools/rocprof_family_ms.py "$O/${P}_kernel_stats.csv" "$O/kernel_ms_per_step.json" > /dev/null
mkdir -p profiles/r06 && cp "$O/kernel_ms_per_step.json" profiles/r06/kernel_ms_per_step.json
python3 bench.py --steps 50 --warmup 10 > "$O/${P}_bench.json" 2> "$O/${P}_bench.err"
python3 tools/profile_step.py > "$O/${P}_per_shape_in_step.log" 2>/dev/null
(cd tools && python3 bench_spade_fused.py > "../$O/${P}_spade_fused_microbench.log" 2>/dev/null)
python3 tools/bench_inference.py > "$O/${P}_inference.log" 2>/dev/null
python3 tools/check_cfg5.py > "$O/${P}_cfg5_640x384_bs4.log" 2>/dev/null
python3 tools/trace_copies.py > "$O/${P}_torch_launches_per_step.log" 2>/dev/null
python3 tools/bench_pack.py > "$O/${P}_weight_pack_microbench.log" 2>/dev/null
python3 tools/bench_mm.py > "$O/${P}_hipblaslt_same_gemm_shapes.log" 2>/dev/null
python3 tools/check_wgrad_batch.py --bench 2>&1 | grep -v amdgpu.ids > "$O/${P}_wgrad_batch_microbench.log"
python3 tools/bench_small.py 2>&1 | grep -v amdgpu.ids > "$O/${P}_degenerate_channel_convs_microbench.log"
python3 tools/bench_sn.py 2>&1 | grep " bank " > "$O/${P}_spectral_norm_chain_microbench.log"
python3 tools/check_plane.py --bench 2>&1 | grep -v amdgpu.ids > "$O/${P}_plane_conv_microbench.log"
S2E_WGRAD_FLAT=62 python3 tools/check_wgrad_flat.py --bench --each 2>&1 | grep -v amdgpu.ids > "$O/${P}_wgrad_flat_microbench_all_kinds_on.log"
S2E_WGRAD_FLAT=0 python3 tools/check_wgrad_flat.py --bench --each 2>&1 | grep -v amdgpu.ids > "$O/${P}_wgrad_flat_microbench_generic_side.log"
python3 tools/check_c8.py --bench 2>&1 | grep -v amdgpu.ids > "$O/${P}_c8_first_layer_microbench.log"
S2E_CONV_C8=0 python3 tools/check_c8.py --bench 2>&1 | grep -v amdgpu.ids > "$O/${P}_c8_first_layer_microbench_generic_side.log"
bash tools/kt_wgrad.sh "0 48" > "$O/${P}_wgrad_launches_in_step_flat_off_on.log" 2>&1
tail -c 1500 "$O/${P}_bench.json"
