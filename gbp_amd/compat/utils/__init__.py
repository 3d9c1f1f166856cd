from . import lie_algebra
from . import transformations
from . import read_balfile
from . import gaussian
from . import derivatives
