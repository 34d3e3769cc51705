"""Robot action tuples accepted by SocialNavGym.step (reference: crowd_nav/utils/action.py:1-7)."""
from collections import namedtuple

ActionXY = namedtuple("ActionXY", ["vx", "vy"])            # holonomic
ActionRot = namedtuple("ActionRot", ["v", "r"])            # unicycle
ActionXYW = namedtuple("ActionXYW", ["bvx", "bvy", "w"])   # holonomic3 (body-frame velocity + yaw rate)
NewState = namedtuple("NewState", ["px", "py", "vx", "vy"])
NewHeadedState = namedtuple("NewHeadedState", ["px", "py", "theta", "bvx", "bvy", "w"])
