# A/B of two builds of libghm.so inside ONE gpurun call (same box, same clocks): build the baseline, copy it to
# gan_heightmaps_amd/libghm_old.so, build the candidate, copy it to libghm_new.so, then
#   gpurun -- 'bash tools/ab_libs.sh "<conv_bench geometry>" <kinds> [more conv_bench args]'
cd gan_heightmaps_amd
for v in old new old new; do
  cp libghm_$v.so libghm.so
  echo "== $v"
  (cd ..; python tools/conv_bench.py $1 --kinds $2 "${@:3}")
done
