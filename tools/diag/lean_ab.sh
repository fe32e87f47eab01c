for lean in 0 1; do
  echo "== E2E_F1_LEAN=$lean"
  for dt in f32 bf16; do E2E_F1_LEAN=$lean python tools/diag/wide_time.py $dt 2>&1 | tail -1; done
  for shape in "512 256 65 64" "256 1000 29 100" "256 1000 29 127" "256 1000 29 60" "256 500 48 110" "512 256 29 64"; do
    echo -n "shape $shape: "; E2E_F1_LEAN=$lean python tools/diag/time_shape.py $shape 20 2>&1 | tail -1
  done
done
