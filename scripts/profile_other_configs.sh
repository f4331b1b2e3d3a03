set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06p
bash scripts/profile_any.sh n256 "--n 256 --steps 4 --warmup 1 --no-cpu-baseline --no-host-callback --survey-steps 0" HBM MFMA > gpurun_out/r06p/n256.log 2>&1
python scripts/pmc_summary2.py gpurun_out/prof_n256 gpurun_out/r06p n256 >> gpurun_out/r06p/n256.log 2>&1
echo n256 done
bash scripts/profile_any.sh cfg2 "--config cfg2 --steps 20 --warmup 3 --no-cpu-baseline" HBM SQ1 SQ2 VALU > gpurun_out/r06p/cfg2.log 2>&1
python scripts/pmc_summary2.py gpurun_out/prof_cfg2 gpurun_out/r06p cfg2 >> gpurun_out/r06p/cfg2.log 2>&1
echo cfg2 done
bash scripts/profile_any.sh cfg5 "--config cfg5 --no-cpu-baseline --cfg5-replicas 1" VALU SQ1   # replicas 1: every launch of the run is a 4096-fit launch (per-launch counter averages) > gpurun_out/r06p/cfg5.log 2>&1
python scripts/pmc_summary2.py gpurun_out/prof_cfg5 gpurun_out/r06p cfg5 >> gpurun_out/r06p/cfg5.log 2>&1
echo cfg5 done
ls gpurun_out/r06p
