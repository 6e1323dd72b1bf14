# needs a second build without the parallel merge marking beside the product:
#   make -C moped_amd/csrc EXTRA=-DMS_NO_PAR BUILD=build_nopar OUT=../libmoped_hip_nopar.so
set -e
for rep in 1 2; do
for lib in moped_amd/libmoped_hip.so moped_amd/libmoped_hip_nopar.so; do
  echo "== $lib"
  MH_LIB_PATH=$PWD/$lib timeout -k 10 200 python scripts/image_frame_bench.py 20 4 3000 2>&1 | grep -v amdgpu.ids
  MH_LIB_PATH=$PWD/$lib timeout -k 10 200 python scripts/image_frame_bench.py 20 16 3000 2>&1 | grep "image->"
done
done
for lib in moped_amd/libmoped_hip.so moped_amd/libmoped_hip_nopar.so; do
  echo "== bench $lib"
  MH_LIB_PATH=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config'].get('objects_per_frame'))"
done
