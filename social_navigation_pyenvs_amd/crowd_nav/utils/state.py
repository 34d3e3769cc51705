"""Observation / state records handed to CrowdNav policies (reference: crowd_nav/utils/state.py:1-77).

Same field names, ``position`` / ``velocity`` tuples, ``__add__`` (tuple concatenation used by the
policies to flatten a joint state) and ``__str__`` as the reference classes, so policies written
against the reference consume them unchanged."""


class _Record:
    _fields = ()

    def _values(self):
        return tuple(getattr(self, f) for f in self._fields)

    def __add__(self, other):
        return other + self._values()

    def __str__(self):
        return " ".join(str(x) for x in self._values())


class ObservableState(_Record):
    _fields = ("px", "py", "vx", "vy", "radius")

    def __init__(self, px, py, vx, vy, radius):
        self.px, self.py, self.vx, self.vy, self.radius = px, py, vx, vy, radius
        self.position = (px, py)
        self.velocity = (vx, vy)


class ObservableStateHeaded(_Record):
    _fields = ("px", "py", "vx", "vy", "radius", "theta", "omega")

    def __init__(self, px, py, vx, vy, radius, theta, omega):
        self.px, self.py, self.vx, self.vy, self.radius = px, py, vx, vy, radius
        self.theta, self.omega = theta, omega
        self.position = (px, py)
        self.velocity = (vx, vy)


class FullState(_Record):
    _fields = ("px", "py", "vx", "vy", "radius", "gx", "gy", "v_pref", "theta")

    def __init__(self, px, py, vx, vy, radius, gx, gy, v_pref, theta):
        self.px, self.py, self.vx, self.vy, self.radius = px, py, vx, vy, radius
        self.gx, self.gy, self.v_pref, self.theta = gx, gy, v_pref, theta
        self.headed = False
        self.position = (px, py)
        self.goal_position = (gx, gy)
        self.velocity = (vx, vy)


class FullStateHeaded(FullState):
    _fields = FullState._fields + ("w",)

    def __init__(self, px, py, vx, vy, radius, gx, gy, v_pref, theta, w):
        super().__init__(px, py, vx, vy, radius, gx, gy, v_pref, theta)  # velocity = body velocity
        self.w = w
        self.headed = True


class JointState:
    def __init__(self, self_state, human_states):
        assert isinstance(self_state, FullState)
        for human_state in human_states:
            assert isinstance(human_state, (ObservableState, ObservableStateHeaded))
        self.self_state = self_state
        self.human_states = human_states
