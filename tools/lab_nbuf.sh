for n in 5 4 3; do
  echo "== NBUF $n"
  for geo in "32 32 65 64 64 9" "32 32 65 32 64 9" "32 16 32 128 128 9" "32 16 32 64 128 9" "32 8 16 256 256 9" "32 8 16 128 256 9" "32 1 128 256 128 5"; do
    timeout -k 5 60 tools/igemm_lab_n$n $geo | tail -1 || exit 1
  done
done
