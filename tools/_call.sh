set -e
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r5 2>&1 | tail -12
