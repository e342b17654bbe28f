#!/bin/bash
# Builds the probe executables the scratch/run*.sh and collect_r0*.sh scripts run, from their sources in scratch/ (they are not
# tracked: ADVICE r4).  Run in the build container (hipcc cross-compiles); the binaries travel to the GPU box with gpurun.
set -e
cd "$(dirname "$0")/.."
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off"
$H -o scratch/chain_probe scratch/chain_probe.hip
$H -o scratch/dpp_rate scratch/dpp_rate.hip
$H -DGPT_PD_STAMPS -o scratch/potf2_stamps scratch/potf2_stamps.hip
$H -DGPT_PD_STAMPS -o scratch/potf2_la_stamps scratch/potf2_la_stamps.hip
$H -o scratch/potf2_la_events scratch/potf2_la_events.hip
$H -DGPT_PD_STAMPS -DGPT_PD_TRACE -o scratch/potf2_la_trace scratch/potf2_la_stamps.hip
$H -DGPT_PU_STAMPS -o scratch/r05_upd_stamps scratch/r05_upd_stamps.hip
ls -la scratch/chain_probe scratch/dpp_rate scratch/potf2_stamps scratch/potf2_la_stamps scratch/potf2_la_events scratch/potf2_la_trace scratch/r05_upd_stamps
