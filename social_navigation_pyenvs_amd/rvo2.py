"""``rvo2`` -- drop-in for the Python-RVO2 module the reference imports, backed by the ORCA HIP kernel.

The reference reaches RVO2 through the ``rvo2.PyRVOSimulator`` object API (second seam):
  /root/reference/social_gym/src/motion_model_manager.py:8       ``import rvo2``
  :237-246   PyRVOSimulator(...), addAgent(...), addObstacle(...), processObstacles()
  :105-141   setAgentPosition / setAgentVelocity / setAgentPrefVelocity, :157-158 setAgentRadius
  :386-394   setTimeStep(dt); doStep(); getAgentPosition / getAgentVelocity
  /root/reference/crowd_nav/policy_no_train/orca.py:95-129       one simulator per decision, robot = agent 0
``import social_navigation_pyenvs_amd.rvo2 as rvo2`` keeps those call sites unchanged.  ``doStep`` is ONE launch of
``cs_step`` (type CS_ORCA, one world) on the agents' current rows: neighbour search, ORCA half-planes, the 2-D linear
programmes and ``position += velocity * timeStep`` (RVO2 ``Agent::computeNeighbors / computeNewVelocity / update``,
restated in csrc/orca.hip; parity with the third-party library itself is unpinned).

Supported: any number of agents (above 512, or when a world's obstacle / neighbour columns outgrow a block's LDS, the neighbours come
from a uniform grid in HBM: csrc/orca.hip k_bw_orca_step), per-agent radius / maxSpeed / position / velocity / preferred velocity,
and per-agent ``neighborDist`` / ``maxNeighbors`` (<= 16) / ``timeHorizon`` / ``timeHorizonObst`` as RVO2 keeps them
(``addAgent``'s arguments, ``setAgent*``; ``cs_worlds.d_orca_agent_params``) -- the reference itself passes ORCA_DEFAULTS for every
agent (motion_model_manager.py:14, :241), which keeps the register-resident ten-neighbour solve.  Static obstacles: ``addObstacle(vertices)`` (counter-clockwise
polygons, or two vertices for a one-sided wall) + ``processObstacles()`` build RVO2's vertex records (point, unit direction
to the next vertex, convexity, links); the kernel restates the obstacle ORCA lines and the hard-constraint form of
linearProgram3.  ``processObstacles()`` also performs the edge SPLITTING of RVO2's obstacle kd-tree
(``split_obstacles_kdtree``: the pieces are what agents meet as obstacle neighbours); each agent keeps its 16 nearest edges.
"""
from __future__ import annotations

import numpy as np

from .batched import CrowdWorlds

_FAR = 1.0e9  # goal the kernel's own goal / preferred-velocity glue never reaches: the shim owns prefVelocity


def process_obstacle(vertices) -> np.ndarray:
    """RVOSimulator::addObstacle for one polygon: [n, 8] float32 records  point.x, point.y, unitDir.x, unitDir.y, isConvex,
    next, prev, 0  (polygon-local indices).  unitDir points to the next vertex; a vertex is convex when
    leftOf(prev, this, next) >= 0; with two vertices both are convex."""
    v = np.asarray(vertices, dtype=np.float32).reshape(-1, 2)
    n = len(v)
    if n < 2:
        raise ValueError("an obstacle needs at least two vertices")
    out = np.zeros((n, 8), np.float32)
    for i in range(n):
        nxt, prv = (i + 1) % n, (i - 1) % n
        d = v[nxt] - v[i]
        u = d / np.float32(np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1])))
        convex = True
        if n > 2:
            a, b, c = v[prv], v[i], v[nxt]
            convex = np.float32((a[0] - c[0]) * (b[1] - a[1])) - np.float32((a[1] - c[1]) * (b[0] - a[0])) >= 0
        out[i] = [v[i, 0], v[i, 1], u[0], u[1], 1.0 if convex else 0.0, nxt, prv, 0.0]
    return out


def split_obstacles_kdtree(records: np.ndarray) -> np.ndarray:
    """RVOSimulator::processObstacles = KdTree::buildObstacleTree (RVO2 v2.0.x KdTree.cpp buildObstacleTreeRecursive): while it builds the obstacle
    kd-tree RVO2 SPLITS every edge that straddles the line of a node's splitting edge -- a new vertex at the intersection (convex, the direction
    of the edge it cuts, linked between the edge's two ends) -- so the edges an agent later meets as obstacle neighbours are those PIECES.  The
    pieces of an edge give the same ORCA half-planes as the edge except where a piece boundary decides which end vertex defines the velocity
    obstacle.  Restated from the published algorithm in float32 like RVO2 (the library is absent: parity unpinned); the tree itself is
    only a search structure -- the kernels find the pieces by brute force / the uniform grid, same neighbour set.
    `records` [n, 8] as process_obstacle builds them with GLOBAL next / prev; returns [n + splits, 8]."""
    f = np.float32
    EPS = f(1e-5)
    rec = [list(map(float, r)) for r in np.asarray(records, dtype=np.float32)]

    def pt(i):
        return f(rec[i][0]), f(rec[i][1])

    def left_of(a, b, c):                       # det(a - c, b - a)
        return f(f(f(a[0] - c[0]) * f(b[1] - a[1])) - f(f(a[1] - c[1]) * f(b[0] - a[0])))

    def det(ax, ay, bx, by):
        return f(f(ax * by) - f(ay * bx))

    def build(obst):
        if not obst:
            return
        n = len(obst)
        best, split = (n, n), 0
        for i in range(n):
            i1 = obst[i]; i2 = int(rec[i1][5])
            left = right = 0
            for j in range(n):
                if i == j:
                    continue
                j1 = obst[j]; j2 = int(rec[j1][5])
                a = left_of(pt(i1), pt(i2), pt(j1)); b = left_of(pt(i1), pt(i2), pt(j2))
                if a >= -EPS and b >= -EPS:
                    left += 1
                elif a <= EPS and b <= EPS:
                    right += 1
                else:
                    left += 1; right += 1
            cand = (max(left, right), min(left, right))
            if cand < (max(best), min(best)):
                best, split = (left, right), i
        i1 = obst[split]; i2 = int(rec[i1][5])
        lefts, rights = [], []
        for j in range(n):
            if j == split:
                continue
            j1 = obst[j]; j2 = int(rec[j1][5])
            a = left_of(pt(i1), pt(i2), pt(j1)); b = left_of(pt(i1), pt(i2), pt(j2))
            if a >= -EPS and b >= -EPS:
                lefts.append(j1)
            elif a <= EPS and b <= EPS:
                rights.append(j1)
            else:                                # the edge j1 -> j2 straddles the line of the splitting edge: cut it there
                p1, p2, q1, q2 = pt(i1), pt(i2), pt(j1), pt(j2)
                t = f(det(f(p2[0] - p1[0]), f(p2[1] - p1[1]), f(q1[0] - p1[0]), f(q1[1] - p1[1])) /
                      det(f(p2[0] - p1[0]), f(p2[1] - p1[1]), f(q1[0] - q2[0]), f(q1[1] - q2[1])))
                sx, sy = f(q1[0] + f(t * f(q2[0] - q1[0]))), f(q1[1] + f(t * f(q2[1] - q1[1])))
                new = len(rec)
                rec.append([float(sx), float(sy), rec[j1][2], rec[j1][3], 1.0, float(j2), float(j1), 0.0])
                rec[j1][5] = float(new); rec[j2][6] = float(new)
                if a > 0:
                    lefts.append(j1); rights.append(new)
                else:
                    rights.append(j1); lefts.append(new)
        build(lefts)
        build(rights)

    build(list(range(len(rec))))
    return np.asarray(rec, dtype=np.float32).reshape(-1, 8)


def process_obstacles(polygons, kdtree_split=True) -> np.ndarray:
    """All polygons of a scene in one vertex table (next / prev become table indices), then RVO2's processObstacles(): the edge splitting of its
    obstacle kd-tree (split_obstacles_kdtree; ``kdtree_split=False``: the polygons' own edges, what rounds 3-5 stepped)."""
    recs, base = [], 0
    for poly in polygons:
        r = process_obstacle(poly)
        r[:, 5] += base
        r[:, 6] += base
        base += len(r)
        recs.append(r)
    out = np.concatenate(recs) if recs else np.zeros((0, 8), np.float32)
    return split_obstacles_kdtree(out) if (kdtree_split and len(out)) else out


class PyRVOSimulator:
    def __init__(self, timeStep, neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, radius, maxSpeed, velocity=(0, 0)):
        self._dt = float(timeStep)
        self._defaults = dict(neighborDist=float(neighborDist), maxNeighbors=int(maxNeighbors), timeHorizon=float(timeHorizon),
                              timeHorizonObst=float(timeHorizonObst), radius=float(radius), maxSpeed=float(maxSpeed),
                              velocity=(float(velocity[0]), float(velocity[1])))
        self._pos, self._vel, self._pref, self._radius, self._maxspeed = [], [], [], [], []
        self._agent = []     # per agent: [neighborDist, maxNeighbors, timeHorizon, timeHorizonObst] (RVO2 keeps the four per agent)
        self._time = 0.0
        self._cw = None
        self._polygons, self._vertices = [], None

    # ------------------------------------------------------------------ building the scene
    def setAgentDefaults(self, neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, radius, maxSpeed, velocity=(0, 0)):
        # (the defaults of agents added from now on: agents already in the simulator keep theirs, as in RVO2)
        self._defaults.update(neighborDist=float(neighborDist), maxNeighbors=int(maxNeighbors), timeHorizon=float(timeHorizon),
                              timeHorizonObst=float(timeHorizonObst), radius=float(radius), maxSpeed=float(maxSpeed),
                              velocity=(float(velocity[0]), float(velocity[1])))

    def addAgent(self, pos, neighborDist=None, maxNeighbors=None, timeHorizon=None, timeHorizonObst=None, radius=None,
                 maxSpeed=None, velocity=None):
        d = self._defaults
        pick = lambda val, name: float(d[name] if val is None else val)
        mn = int(pick(maxNeighbors, "maxNeighbors"))
        if not 0 <= mn <= 16:
            raise ValueError("maxNeighbors must be in 0..16")
        self._agent.append([pick(neighborDist, "neighborDist"), float(mn), pick(timeHorizon, "timeHorizon"), pick(timeHorizonObst, "timeHorizonObst")])
        self._pos.append([float(pos[0]), float(pos[1])])
        v = d["velocity"] if velocity is None else velocity
        self._vel.append([float(v[0]), float(v[1])])
        self._pref.append([0.0, 0.0])
        self._radius.append(float(d["radius"] if radius is None else radius))
        self._maxspeed.append(float(d["maxSpeed"] if maxSpeed is None else maxSpeed))
        self._cw = None
        return len(self._pos) - 1

    def addObstacle(self, vertices):
        """Counter-clockwise polygon (or two vertices: a wall seen from its right side).  Returns the number of the
        obstacle's first vertex, like RVO2."""
        first = sum(len(p) for p in self._polygons)
        self._polygons.append([(float(v[0]), float(v[1])) for v in vertices])
        return first

    def processObstacles(self):
        """Builds the vertex table the kernel reads (RVO2 builds its obstacle kd-tree here)."""
        self._vertices = process_obstacles(self._polygons) if self._polygons else None
        self._cw = None

    # ------------------------------------------------------------------ stepping
    def setTimeStep(self, timeStep):
        self._dt = float(timeStep)

    def getTimeStep(self):
        return self._dt

    def getGlobalTime(self):
        return self._time

    def doStep(self):
        n = len(self._pos)
        if n == 0:
            self._time += self._dt
            return
        S = np.zeros((1, n, 13), np.float32)
        S[0, :, 0:2] = self._pos
        S[0, :, 3:5] = self._vel
        S[0, :, 5:7] = self._pref
        S[0, :, 8] = self._radius
        S[0, :, 9] = 1.0
        S[0, :, 10:12] = _FAR
        S[0, :, 12] = self._maxspeed
        if self._cw is None or self._cw.n != n:
            if self._polygons and self._vertices is None:
                raise RuntimeError("processObstacles() has to be called after addObstacle() (RVO2 ignores unprocessed obstacles)")
            self._cw = CrowdWorlds(S, np.full((1, n, 1, 2), _FAR, np.float32), None, None, None, type="orca",
                                   orca_vertices=self._vertices)
            self._params_dirty = True
        else:
            self._cw.set_states(S)
        if getattr(self, "_params_dirty", True):
            ap = np.asarray(self._agent, np.float32).reshape(1, n, 4)
            first = ap[0, 0]
            # one parameter set for everybody (what the reference builds: ORCA_DEFAULTS for every addAgent): the scalar descriptor
            # fields and the register-resident solve; otherwise the per-agent array (cs_worlds.d_orca_agent_params)
            self._cw.orca_params = dict(neighbor_dist=float(ap[0, :, 0].max()), max_neighbors=int(ap[0, :, 1].max()),
                                        time_horizon=float(first[2]), time_horizon_obst=float(first[3]))
            self._cw.set_orca_agent_params(None if np.all(ap[0] == first) else ap)
            self._params_dirty = False
        self._cw.step(self._dt, 1, None)
        out = self._cw.get_states()[0]
        self._pos = out[:, 0:2].astype(np.float64).tolist()
        self._vel = out[:, 3:5].astype(np.float64).tolist()
        self._time += self._dt

    # ------------------------------------------------------------------ accessors (rvo2.pyx names)
    def getNumAgents(self):
        return len(self._pos)

    def getAgentPosition(self, i):
        return (self._pos[i][0], self._pos[i][1])

    def getAgentVelocity(self, i):
        return (self._vel[i][0], self._vel[i][1])

    def getAgentPrefVelocity(self, i):
        return (self._pref[i][0], self._pref[i][1])

    def getAgentRadius(self, i):
        return self._radius[i]

    def getAgentMaxSpeed(self, i):
        return self._maxspeed[i]

    def getAgentNeighborDist(self, i):
        return self._agent[i][0]

    def getAgentMaxNeighbors(self, i):
        return int(self._agent[i][1])

    def getAgentTimeHorizon(self, i):
        return self._agent[i][2]

    def getAgentTimeHorizonObst(self, i):
        return self._agent[i][3]

    def getNumObstacleVertices(self):
        return sum(len(p) for p in self._polygons)

    def getObstacleVertex(self, i):
        flat = [v for p in self._polygons for v in p]
        return flat[i]

    def setAgentPosition(self, i, pos):
        self._pos[i] = [float(pos[0]), float(pos[1])]

    def setAgentVelocity(self, i, vel):
        self._vel[i] = [float(vel[0]), float(vel[1])]

    def setAgentPrefVelocity(self, i, vel):
        self._pref[i] = [float(vel[0]), float(vel[1])]

    def setAgentRadius(self, i, radius):
        self._radius[i] = float(radius)

    def setAgentMaxSpeed(self, i, maxSpeed):
        self._maxspeed[i] = float(maxSpeed)

    def _set_agent(self, i, col, v):
        self._agent[i][col] = float(v)
        self._params_dirty = True

    def setAgentNeighborDist(self, i, v):
        self._set_agent(i, 0, v)

    def setAgentMaxNeighbors(self, i, v):
        if not 0 <= int(v) <= 16:
            raise ValueError("maxNeighbors must be in 0..16")
        self._set_agent(i, 1, int(v))

    def setAgentTimeHorizon(self, i, v):
        self._set_agent(i, 2, v)

    def setAgentTimeHorizonObst(self, i, v):
        self._set_agent(i, 3, v)
