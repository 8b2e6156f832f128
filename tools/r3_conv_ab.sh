for v in "" _enostore _enogn; do
  echo "variant librald_hip$v.so"; RALD_LIB_OVERRIDE=rald_amd/librald_hip$v.so timeout -k 10 200 python tools/time_enc.py 2>&1 | grep "B=8"
done
