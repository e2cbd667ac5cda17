"""Type aliases used by the reference under TYPE_CHECKING only."""
from typing import Any, Dict

DataArrayDict = Dict[str, Any]
NDArrayLike = Any
NDArrayLikeDict = Dict[str, Any]
PropertyDict = Dict[str, Dict[str, Any]]
