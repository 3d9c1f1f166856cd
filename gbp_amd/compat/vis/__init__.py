"""Headless stand-in for the reference's trimesh/pyglet viewer package (same module names, no GUI dependencies)."""
from . import ba_vis, vis_scene

__all__ = ['ba_vis', 'vis_scene']
