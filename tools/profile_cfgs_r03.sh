#!/bin/bash
# kernel trace + HBM traffic (FETCH_SIZE, WRITE_SIZE in runs of their own) of the other BASELINE configurations at full size
export LIGHT=1
bash tools/profile_r02.sh r03_cfg2 --order 18 --channels 64 --periods 1 || exit 1
bash tools/profile_r02.sh r03_cfg5 --order 20 --channels 1024 --periods 1 || exit 1
bash tools/profile_r02.sh r03_cfg5f32 --order 20 --channels 1024 --periods 1 --dtype f32 || exit 1
bash tools/profile_r02.sh r03_cfg4 --order 24 --channels 256 --periods 1 || exit 1
