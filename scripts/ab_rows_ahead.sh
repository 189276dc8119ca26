cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
echo "rows ahead"; bash scripts/ab_loop.sh --logging_nan_inf_filter false
echo "rows per micro-step"; SD_ROWS_AHEAD=0 bash scripts/ab_loop.sh --logging_nan_inf_filter false
done
