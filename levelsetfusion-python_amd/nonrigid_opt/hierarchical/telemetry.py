"""Per-iteration telemetry of the hierarchical optimizer and its .npz persistence (reference:
nonrigid_opt/hierarchical/hierarchical_optimization_visualizer.py:154-250 -- the data classes and the save / load of
telemetry_log.npz; the video rendering around them is out of scope)."""
import os

import numpy as np


class OptimizationIterationData:
    def __init__(self, warp_fields, data_term_gradients, tikhonov_term_gradients):
        self.warp_fields = warp_fields
        self.data_term_gradients = data_term_gradients
        self.tikhonov_term_gradients = tikhonov_term_gradients

    def get_warp_fields(self):
        return self.warp_fields

    def get_data_term_gradients(self):
        return self.data_term_gradients

    def get_tikhonov_term_gradients(self):
        return self.tikhonov_term_gradients

    def get_frame_count(self):
        return len(self.warp_fields)


def _stack(fields):
    return np.concatenate(fields, axis=-1) if len(fields) else np.array([])


def save_telemetry_log(telemetry_log, output_folder):
    """telemetry_log.npz with l{i}_warp_fields / l{i}_data_term_gradients / l{i}_tikhonov_term_gradients, every
    entry the per-iteration vector fields stacked along the last axis (…visualizer.py:210-228)"""
    os.makedirs(output_folder, exist_ok=True)
    entries = {}
    for i, level in enumerate(telemetry_log):
        entries["l{:d}_warp_fields".format(i)] = _stack(level.get_warp_fields())
        entries["l{:d}_data_term_gradients".format(i)] = _stack(level.get_data_term_gradients())
        entries["l{:d}_tikhonov_term_gradients".format(i)] = _stack(level.get_tikhonov_term_gradients())
    path = os.path.join(output_folder, "telemetry_log.npz")
    np.savez_compressed(path, **entries)
    return path


def load_telemetry_log(output_folder, components=2):
    """inverse of save_telemetry_log (…visualizer.py:231-250)"""
    archive = np.load(os.path.join(output_folder, "telemetry_log.npz"))
    log = []
    for i in range(len(archive.files) // 3):
        def split(name):
            a = archive["l{:d}_{:s}".format(i, name)]
            return [] if a.ndim < 3 else np.split(a, a.shape[-1] // components, axis=-1)
        log.append(OptimizationIterationData(split("warp_fields"), split("data_term_gradients"),
                                             split("tikhonov_term_gradients")))
    return log
