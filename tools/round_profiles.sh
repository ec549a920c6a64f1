#!/bin/bash
# usage (GPU box, via gpurun): bash tools/round_profiles.sh <prefix>   e.g. round4_v1
# kernel-trace stats + the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) for the bench shape and
# the shapes bench.py lists under `secondary`:
#   <prefix>            collab-like uniform (BASELINE configs[1], the driver's shape)
#   <prefix>_ppa        ppa-like uniform (configs[3]; MinHash table 295 MB: HBM-resident)
#   <prefix>_citation2  citation2-like uniform, h = 3 (configs[4]; 1.5 GB MinHash table: HBM-resident random gathers)
#   <prefix>_powerlaw   collab-like, endpoint weights ~ rank^-0.5 (hub rows); _powerlaw09: rank^-0.9 (mega rows, dense CSR buckets)
#   <prefix>_ppa_powerlaw / _citation2_powerlaw   the power-law generator at configs[3] / [4] size (rank^-0.5)
#   <prefix>_elph / _buddy   the ELPH call sequence at B = 2048 and the BUDDY precompute at collab size
# then, in the build container: bash tools/round_profiles_summarise.sh <prefix> (copies the summaries into profiles/ and
# records the per-shape fabric bytes in profiles/pmc_traffic.json)
P=$1
R=$GRAFT_REPO_ROOT
bash $R/tools/prof.sh $P
bash $R/tools/prof.sh ${P}_ppa --config ppa --steps 10 --warmup 2
bash $R/tools/prof.sh ${P}_citation2 --config citation2 --steps 5 --warmup 2
bash $R/tools/prof.sh ${P}_powerlaw --graph powerlaw --alpha 0.5
bash $R/tools/prof.sh ${P}_powerlaw09 --graph powerlaw --alpha 0.9
bash $R/tools/prof.sh ${P}_ppa_powerlaw --config ppa --graph powerlaw --alpha 0.5 --steps 10 --warmup 2
bash $R/tools/prof.sh ${P}_citation2_powerlaw --config citation2 --graph powerlaw --alpha 0.5 --steps 5 --warmup 2
bash $R/tools/prof.sh ${P}_elph --api elph --batch 2048
bash $R/tools/prof.sh ${P}_buddy --api buddy --steps 20 --warmup 3
