#!/bin/bash
# usage (build container, after gpurun merged gpurun_out/prof_<prefix>*): bash tools/round_profiles_summarise.sh <prefix>
P=$1
cd "$(dirname "$0")/.."
python tools/summarise_prof.py $P --set-traffic --shape collab_uniform > /dev/null
python tools/summarise_prof.py ${P}_ppa --shape ppa_uniform > /dev/null
python tools/summarise_prof.py ${P}_citation2 --shape citation2_uniform > /dev/null
python tools/summarise_prof.py ${P}_powerlaw --shape collab_powerlaw05 > /dev/null
python tools/summarise_prof.py ${P}_powerlaw09 --shape collab_powerlaw09 > /dev/null
python tools/summarise_prof.py ${P}_ppa_powerlaw --shape ppa_powerlaw05 > /dev/null
python tools/summarise_prof.py ${P}_citation2_powerlaw --shape citation2_powerlaw05 > /dev/null
python tools/summarise_prof.py ${P}_elph --shape collab_uniform_elph > /dev/null
python tools/summarise_prof.py ${P}_buddy --shape collab_uniform_buddy > /dev/null
python tools/roofline_table.py ${P} collab > /dev/null
python tools/roofline_table.py ${P}_ppa ppa > /dev/null
python tools/roofline_table.py ${P}_citation2 citation2 > /dev/null
python tools/roofline_table.py ${P}_powerlaw collab --graph powerlaw --alpha 0.5 > /dev/null
python tools/roofline_table.py ${P}_powerlaw09 collab --graph powerlaw --alpha 0.9 > /dev/null
python tools/roofline_table.py ${P}_ppa_powerlaw ppa --graph powerlaw --alpha 0.5 > /dev/null
python tools/roofline_table.py ${P}_citation2_powerlaw citation2 --graph powerlaw --alpha 0.5 > /dev/null
ls profiles | grep "^$P" | wc -l
