# needs the previous POSE kernel as moped_amd/libmoped_hip_oldpose.so (build pose.hip of the commit before into a copy of the library)
for rep in 1 2; do for lib in libmoped_hip.so libmoped_hip_oldpose.so; do
  MH_LIB_PATH=$PWD/moped_amd/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --h2d-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib config 1', d['value'], d['config']['objects_per_frame'], flush=True)"
done; done
