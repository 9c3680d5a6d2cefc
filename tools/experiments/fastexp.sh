cd /root/repo
for lib in libdca_hip.so libdca_hip_fastexp.so; do
  echo "== $lib"
  DCA_LIB_PATH=$PWD/pydca_amd/lib/$lib python tools/time_eval.py --L 150 --N 200000 --q 5 --seed 12347 --reps 10
  DCA_LIB_PATH=$PWD/pydca_amd/lib/$lib python tools/time_eval.py --L 200 --N 10000 --q 21 --seed 12345 --reps 10
  DCA_LIB_PATH=$PWD/pydca_amd/lib/$lib python tools/time_eval.py --L 500 --N 50000 --q 21 --seed 12346 --reps 5
done
