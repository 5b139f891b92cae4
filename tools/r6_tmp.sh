cd /root/repo
for r in 1 2 3 4; do
  echo -n "old: "; CINDM_LIB_VARIANT=exp0 python tools/ab1d.py tune 0 0 600 cfg2 2>/dev/null | grep us/step | tail -1
  echo -n "new: "; python tools/ab1d.py tune 0 0 600 cfg2 2>/dev/null | grep us/step | tail -1
done
