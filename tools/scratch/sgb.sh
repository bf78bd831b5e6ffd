cp gan_heightmaps_amd/libghm.so /tmp/new.so
for v in new wg1 wg2; do
if [ $v = new ]; then cp /tmp/new.so gan_heightmaps_amd/libghm.so; else cp tools/scratch/libghm_$v.so gan_heightmaps_amd/libghm.so; fi
echo $v
for g in "8 64 256 256 128 5 1 2" "4 512 128 128 128 3 1 1" "4 256 256 256 64 3 1 1" "8 128 128 128 256 3 2 1"; do
  python tools/conv_bench.py $g --kinds wgrad 2>&1 | grep -v "^$"
done
for r in 1 2; do python bench.py --steps 30 --warmup 5 2>&1 | tail -1 | cut -c55-75; done
done
cp /tmp/new.so gan_heightmaps_amd/libghm.so
python -m pytest tests/test_gpu_ops.py -x -q -k conv 2>&1 | grep -E "passed|failed"
