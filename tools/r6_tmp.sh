cd /root/repo
for r in 1 2 3; do
  echo -n "conditional touches: "; CINDM_LIB_VARIANT=exp0 python tools/ab1d.py tune 0 0 600 cfg2 2>/dev/null | grep us/step | tail -1
  echo -n "exact count:         "; python tools/ab1d.py tune 0 0 600 cfg2 2>/dev/null | grep us/step | tail -1
done
echo -n "cfg3 conditional: "; CINDM_LIB_VARIANT=exp0 python tools/ab1d.py tune 0 0 300 cfg3 2>/dev/null | grep us/step | tail -1
echo -n "cfg3 exact count: "; python tools/ab1d.py tune 0 0 300 cfg3 2>/dev/null | grep us/step | tail -1
echo -n "cfg3 conditional: "; CINDM_LIB_VARIANT=exp0 python tools/ab1d.py tune 0 0 300 cfg3 2>/dev/null | grep us/step | tail -1
echo -n "cfg3 exact count: "; python tools/ab1d.py tune 0 0 300 cfg3 2>/dev/null | grep us/step | tail -1
