"""Headless stand-in for vis/vis_scene.py (trimesh/pyglet GUI; out of scope)."""


def view(cam_params, landmarks, K, fov=(640, 480)):
    return None


def view_from_graph(graph, fov=(640, 480)):
    return None
