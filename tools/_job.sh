cd $GRAFT_REPO_ROOT
python3 tools/ablate.py run mdpp_image.hip cfg4 numpy shipped os_straight shipped os_straight 2>&1 | cut -c1-200
