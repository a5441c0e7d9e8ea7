cd $GRAFT_REPO_ROOT
for k in 1 2; do
for l in r5 5f96f84 6d0ef33 780098b 88070a1 HEAD; do
  if [ $l = HEAD ]; then P=avian_amd/csrc/libavian_mi355x.so; else P=avian_amd/csrc/ab/libavian_$l.so; fi
  echo -n "$l run $k: "; AVN_AB_OLDER_LIBRARY=1 AVN_LIB_PATH=$PWD/$P python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
done; done
