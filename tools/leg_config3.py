import sys, importlib, json, torch
sys.path.insert(0, '.')
import bench
ion = importlib.import_module("neural-ode-ion-channels_amd")
w, _ = bench.load_weights()
print(json.dumps(bench.config3_leg(ion, torch.device("cuda:0"), w), indent=1))
