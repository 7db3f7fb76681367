"""precoder config C timing (tools only)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_extra as be
print(json.dumps(be.precoder_config_c()))
