# HF Trainer loop: the accumulation window fetched on its own stream + loss rows selected ahead (default) vs SD_ROWS_AHEAD=0
cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "ahead (default)"; bash scripts/ab_loop.sh
echo "SD_ROWS_AHEAD=0"; SD_ROWS_AHEAD=0 bash scripts/ab_loop.sh
echo "ahead, nan filter off"; bash scripts/ab_loop.sh --logging_nan_inf_filter false
echo "SD_ROWS_AHEAD=0, nan filter off"; SD_ROWS_AHEAD=0 bash scripts/ab_loop.sh --logging_nan_inf_filter false
done
