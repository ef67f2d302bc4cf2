"""Result files of the reference's `main`s, under the reference's names, so that its figure scripts keep working
(consumers: confrez/generate_figs.py):

    <rl_file_name>_<agent>_zu0.pkl       warm start on the collocation grid  (confrez/control/vehicle.py:927)
    <rl_file_name>_<agent>_zufinal.pkl   single-vehicle collocation plan     (vehicle.py:928)
    <rl_file_name>_opt.pkl               joint plan, Dict[agent, VehiclePrediction]   (multi_vehicle_planner.py:668)
    <rl_file_name>_follower_final.pkl, _follower_iter_time.pkl   closed loop   (vehicle_follower.py:665-670)

The reference writes them with `dill`; `dill.load` reads plain pickles, and nothing in these objects needs dill's
extensions (dataclasses of numpy arrays and lists), so `pickle` is used when dill is not installed.
"""
import pickle

try:  # the reference's choice when it is there
    import dill as _pk
except ImportError:  # pragma: no cover - this image has no dill
    _pk = pickle


def dump(obj, path: str):
    with open(path, "wb") as f:
        _pk.dump(obj, f)
    return path


def load(path: str):
    with open(path, "rb") as f:
        return _pk.load(f)
