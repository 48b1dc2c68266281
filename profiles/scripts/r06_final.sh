set -x
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r06f
rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 50 --warmup 10 --streams 1 --no-overlap --no-scatter --no-cpu-baseline --no-extras > $O/bench_profiled.json 2> $O/trace.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python bench.py --steps 20 --warmup 5 --streams 1 --no-overlap --no-scatter --no-cpu-baseline --no-extras > $O/pmc_sq.json 2> $O/pmc_sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 20 --warmup 5 --streams 1 --no-overlap --no-scatter --no-cpu-baseline --no-extras > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --steps 20 --warmup 5 --streams 1 --no-overlap --no-scatter --no-cpu-baseline --no-extras > $O/pmc_write.json 2> $O/pmc_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/alleig_trace -- python bench.py --all-eigenvectors --steps 50 --warmup 10 --streams 1 --no-overlap --no-scatter --no-cpu-baseline > $O/alleig_profiled.json 2> $O/alleig_trace.err
python bench.py --workload train --steps 30 --warmup 10 > $O/train.json 2> $O/train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_trace -- python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline > $O/train_profiled.json 2> $O/train_trace.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/train_pmc_sq -- python bench.py --workload train --steps 8 --warmup 3 --no-cpu-baseline > $O/train_pmc.json 2> $O/train_pmc.err
python bench.py --config 0 --steps 50 --warmup 10 --no-scatter > $O/config0.json 2> $O/c0.err
python bench.py --config 2 --steps 50 --warmup 10 --no-scatter > $O/config2.json 2> $O/c2.err
python bench.py --config 4 --steps 20 --warmup 5 > $O/config4.json 2> $O/c4.err
python bench.py --workload dgl --steps 30 --warmup 5 > $O/dgl.json 2> $O/dgl.err
python bench.py --workload evd --steps 30 --warmup 5 > $O/evd.json 2> $O/evd.err
find $O -name "*.csv" | head -40
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/sc_fetch -- python bench.py --workload scatter --steps 5 --warmup 2 > $O/sc_f.json 2> $O/sc_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/sc_write -- python bench.py --workload scatter --steps 5 --warmup 2 > $O/sc_w.json 2> $O/sc_w.err
python bench.py --workload scatter --steps 20 --warmup 5 > $O/scatter.json 2> $O/scatter.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
bash profiles/scripts/pmc_phi_detail.sh > $O/pmc_detail.txt 2>&1
bash profiles/scripts/prof_gnn.sh > $O/gnn_stamps.txt 2>&1
bash profiles/scripts/prof_plan.sh > $O/plan_stamps.txt 2>&1
# round 6 additions: the eigendecomposition's kernel trace + VALU / LDS issue counters, the training step's launch-structure A/B
rocprofv3 --kernel-trace --stats --output-format csv -d $O/evd_trace -- python bench.py --workload evd --steps 30 --warmup 5 --no-cpu-baseline > $O/evd_profiled.json 2> $O/evd_trace.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/evd_pmc -- python bench.py --workload evd --steps 8 --warmup 3 --no-cpu-baseline > $O/evd_pmc.json 2> $O/evd_pmc.err
bash profiles/scripts/train_ab.sh > $O/train_ab.txt 2>&1
python -m signnet_basisnet_amd.build --force > /dev/null 2>&1
# keep what the collector reads (kernel_stats / counter_collection summaries), drop the raw traces: gpurun copies back <= 64 MiB
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
rm -rf gpurun_out/train_ab/*.pt gpurun_out/pmc_detail/p*/
du -sh $O gpurun_out/pmc_detail 2>/dev/null
