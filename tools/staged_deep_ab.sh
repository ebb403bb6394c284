ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  echo "product : $(STAGED_LAB=0 python3 $ROOT/tools/staged_time.py 100000 1024 512 | tail -1)"
  echo "lab     : $(STAGED_LAB=1 python3 $ROOT/tools/staged_time.py 100000 1024 512 | tail -1)"
  echo "lab deep: $(STAGED_LAB=1 MTMC_LAB_LIB=$ROOT/build/var_deep/libmtmc_lab.so python3 $ROOT/tools/staged_time.py 100000 1024 512 | tr '\n' ' ')"
done
