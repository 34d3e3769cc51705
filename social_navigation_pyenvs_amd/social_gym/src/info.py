"""Episode-event records returned in ``info[0]`` (reference: social_gym/src/info.py:1-38)."""


class Timeout:
    def __str__(self):
        return "Timeout"


class ReachGoal:
    def __str__(self):
        return "Reaching goal"


class Danger:
    def __init__(self, min_dist):
        self.min_dist = min_dist

    def __str__(self):
        return "Too close"


class Collision:
    def __str__(self):
        return "Collision"


class Nothing:
    def __str__(self):
        return ""


# order of the info_code the device kernel writes (include/crowdstep.h, cs_collision_reward)
INFO_BY_CODE = (Nothing, Danger, ReachGoal, Collision, Timeout)
