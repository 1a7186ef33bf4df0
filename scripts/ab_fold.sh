cd /root/repo
for r in 1 2 3; do
  for v in 0 4 2; do
    echo "== PD_LIN_FOLD=$v round $r"
    PD_LIN_FOLD=$v python scripts/profile_forward.py 2>/dev/null | grep -E "conv1x1|total ms" | awk '{print}' | head -14
  done
done
