# Copy the results of tools/profile_round.sh (gpurun_out/round/) into profiles/ under this round's names.
#   bash tools/collect_profiles.sh r02
set -e
R=${1:-r02}
SRC=gpurun_out/round
cp $SRC/bench.json profiles/${R}_bench.json
cp $SRC/configs.jsonl profiles/${R}_configs.jsonl
# the bench process is the one with the most kernels in its stats file
big=$(wc -l $(ls -t $SRC/stats/*/*_kernel_stats.csv | head -2) | grep -v total | sort -n | tail -1 | awk '{print $2}')
cp $big profiles/${R}_kernel_stats.csv
cp $(ls -t $SRC/stats_headline/*/*_kernel_stats.csv | head -1) profiles/${R}_headline_kernel_stats.csv
mkdir -p profiles/${R}_pmc
for m in stream auto resident rest; do
  for c in FETCH_SIZE WRITE_SIZE; do
    f=$(ls -t $SRC/pmc/$m/$c/*/*counter_collection.csv | head -1)     # newest: gpurun merges, it does not delete older runs
    cp $f profiles/${R}_pmc/${c}_${m}_counter_collection.csv
  done
done
python tools/pmc_to_json.py $SRC/pmc > /dev/null
echo collected into profiles/${R}_*
