cd $GRAFT_REPO_ROOT
python3 tools/ablate.py run mdpp_image.hip img100_shift numpy nst0 nst10 2>&1 | tail -2
python3 tools/ablate.py run mdpp_image.hip img100_shift numpy image_transforms=rotate nst0 nst10 2>&1 | tail -2
