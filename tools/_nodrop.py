import sys, json, subprocess, os
sys.path.insert(0, '.')
import bench
bench.MODEL_ARGS['DeepFM'] = dict(deep_layers=[256, 128, 64, 32], dropout_fm=0.0, dropout_deep=0.0)
sys.argv = ['bench.py', '--no-cpu-baseline']
bench.main()
