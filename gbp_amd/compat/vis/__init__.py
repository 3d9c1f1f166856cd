from . import vis_scene
from . import ba_vis
