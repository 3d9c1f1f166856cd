"""Headless stand-in for vis/ba_vis.py so that ba.py:8,79-81,103 run unchanged without trimesh / pyglet.

Same three entry points (create_scene, TrimeshSceneViewer.show / .update).  `update` still reads every node's `mu`, what the
real viewer does (vis/ba_vis.py:39-43) -- in one device read for the GPU graph, node by node for any other -- so the data path a
GUI would use stays exercised; the last snapshot is kept in `viewer.cam_params` / `viewer.landmarks`."""
import types

import numpy as np


def create_scene(graph, fov=(640, 480)):
    K = graph.factors[0].args[0] if len(graph.factors) else np.eye(3)
    scene = types.SimpleNamespace()
    scene.camera = types.SimpleNamespace(resolution=np.array(fov), K=K)
    scene.cam_params = [list(c.mu) for c in graph.cam_nodes]
    scene.landmarks = [list(l.mu) for l in graph.lmk_nodes]
    return scene


class TrimeshSceneViewer:
    def __init__(self, scene, resolution=None):
        self.scene, self.resolution = scene, resolution
        self.cam_params, self.landmarks = scene.cam_params, scene.landmarks
        self.n_updates = 0

    def show(self):
        pass

    def update(self, graph):
        if hasattr(graph, '_means'):                        # the device graph: every node's mu in one read (what a per-node loop would assemble)
            cm, lm = graph._means()
            self.cam_params, self.landmarks = cm.tolist(), lm.tolist()
        else:
            self.cam_params = [list(c.mu) for c in graph.cam_nodes]
            self.landmarks = [list(l.mu) for l in graph.lmk_nodes]
        self.n_updates += 1
