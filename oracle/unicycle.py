"""Oracle: deterministic unicycle / Ackermann task functions (test infrastructure).

numpy restatement of the plant, planner, CLF and obstacle CBF of
bayes_cbf/unicycle_move_to_pose.py and bayes_cbf/planner.py (line references per function).
"""
import math
import numpy as np


def normalize_radians(theta):
    """bayes_cbf/misc.py:317-318."""
    return (theta + math.pi) % (2 * math.pi) - math.pi


def angdiff(thetap, theta):
    """unicycle_move_to_pose.py:431-432."""
    return normalize_radians(thetap - theta)


def cartesian2polar(state, state_goal):
    """unicycle_move_to_pose.py:112-139 -> (rho, alpha, beta)."""
    x, y, theta = state
    xg, yg, thetag = state_goal
    xd, yd = xg - x, yg - y
    rho = math.sqrt(xd * xd + yd * yd)
    phi = math.atan2(yd, xd)
    return rho, angdiff(theta, phi), angdiff(thetag, phi)


def ackermann_f(x):
    """AckermannDrive.f_func  (unicycle_move_to_pose.py:222-233)."""
    return np.zeros(3)


def ackermann_g(x, L):
    """AckermannDrive.g_func  (unicycle_move_to_pose.py:235-257): [[cos,0],[sin,0],[0,1/L]]."""
    return np.array([[math.cos(x[2]), 0.0], [math.sin(x[2]), 0.0], [0.0, 1.0 / L]])


def ackermann_step(x, u, dt, L):
    """AckermannDrive.step: explicit Euler  (unicycle_move_to_pose.py:277-282)."""
    return x + (ackermann_f(x) + ackermann_g(x, L) @ u) * dt


class CLFCartesian:
    """unicycle_move_to_pose.py:522-615."""

    def __init__(self, Kp=(0.9, 1.5, 4.0)):
        self.Kp = np.asarray(Kp, dtype=np.float64)

    def clf(self, state, goal):
        rho, alpha, beta = cartesian2polar(state, goal)                       # :527-534
        return (0.5 * self.Kp[0] * rho ** 2 + self.Kp[1] * (1 - math.cos(alpha))
                + self.Kp[2] * (1 - math.cos(beta)))

    def grad_clf(self, state, goal):
        xd, yd, _ = np.asarray(goal) - np.asarray(state)                      # :564-599
        rho, alpha, beta = cartesian2polar(state, goal)
        K = self.Kp
        T = np.array([[-K[0] * xd, -K[1] * math.sin(alpha) * yd / rho ** 2, -K[2] * math.sin(beta) * yd / rho ** 2],
                      [-K[0] * yd, K[1] * math.sin(alpha) * xd / rho ** 2, K[2] * math.sin(beta) * xd / rho ** 2],
                      [0.0, K[1] * math.sin(alpha), 0.0]])
        return T.sum(axis=-1)

    def grad_clf_wrt_goal(self, state, goal):
        xd, yd, _ = np.asarray(goal) - np.asarray(state)                      # :536-562, 601-610
        rho, alpha, beta = cartesian2polar(state, goal)
        K = self.Kp
        T = np.array([[K[0] * xd, K[1] * math.sin(alpha) * yd / rho ** 2, K[2] * math.sin(beta) * yd / rho ** 2],
                      [K[0] * yd, -K[1] * math.sin(alpha) * xd / rho ** 2, -K[2] * math.sin(beta) * xd / rho ** 2],
                      [0.0, 0.0, K[2] * math.sin(beta)]])
        return T.sum(axis=-1)


class ObstacleCBF:
    """unicycle_move_to_pose.py:618-696."""

    def __init__(self, center, radius, term_weights=(0.5, 0.5)):
        self.center = np.asarray(center, dtype=np.float64)
        self.radius = float(radius)
        self.term_weights = tuple(term_weights)

    def cbf(self, s):
        gh = np.asarray(s[:2]) - self.center
        radial = (gh ** 2).sum() - self.radius ** 2                            # :624-625
        ghn = gh / np.linalg.norm(gh)
        heading = math.cos(s[2]) * ghn[0] + math.sin(s[2]) * ghn[1]            # :627-630
        return self.term_weights[0] * radial + self.term_weights[1] * heading

    def grad_cbf(self, s):
        gh = np.asarray(s[:2]) - self.center
        g_rad = np.array([2 * gh[0], 2 * gh[1], 0.0])                           # :642-652
        rho = np.linalg.norm(gh)
        al = math.atan2(gh[1], gh[0])
        th = s[2]
        g_head = np.array([math.sin(al - th) * gh[1] / rho ** 2,               # :654-678
                           -math.sin(al - th) * gh[0] / rho ** 2,
                           -math.sin(th - al)])
        return self.term_weights[0] * g_rad + self.term_weights[1] * g_head


def obstacles_at_mid_from_start_and_goal(x, xg, term_weights=(0.5, 0.5)):
    """unicycle_move_to_pose.py:1562-1570."""
    x = np.asarray(x, dtype=np.float64)
    xg = np.asarray(xg, dtype=np.float64)
    R90 = np.array([[0.0, -1.0], [1.0, 0.0]])
    mid = (x[:2] + xg[:2]) / 2
    off = R90 @ (x[:2] - xg[:2]) / 3
    rad = np.linalg.norm(x[:2] - xg[:2]) / 4
    return [ObstacleCBF(mid + off, rad, term_weights), ObstacleCBF(mid - off, rad, term_weights)]


class PiecewiseLinearPlanner:
    """bayes_cbf/planner.py:19-64."""

    def __init__(self, x0, x_goal, numSteps, dt, frac_time_to_reach_goal=0.7):
        self.x0 = np.asarray(x0, dtype=np.float64)
        self.x_goal = np.asarray(x_goal, dtype=np.float64)
        self.numSteps = numSteps
        self.dt = dt
        xdiff = self.x_goal[:2] - self.x0[:2]
        t2 = min(int(numSteps * frac_time_to_reach_goal), numSteps - 1)
        self._cps = [(t2, np.concatenate([self.x_goal[:2], xdiff / np.linalg.norm(xdiff)])),
                     (numSteps, np.concatenate([self.x_goal[:2], [math.cos(self.x_goal[2]), math.sin(self.x_goal[2])]]))]

    def _interval(self, t):
        prev_t, prev_x = 0, np.concatenate([self.x0[:2], [math.cos(self.x0[2]), math.sin(self.x0[2])]])
        for ct, cx in self._cps:
            if t <= ct:
                break
            prev_t, prev_x = ct, cx
        return (ct, cx), (prev_t, prev_x)

    def _target_step(self, t):
        return min(t + max(int(0.1 * self.numSteps), 1), self.numSteps)

    def plan(self, t):
        t = self._target_step(t)
        (ct, cx), (pt, px) = self._interval(t)
        xp = (cx - px) * (t - pt) / (ct - pt) + px
        return np.array([xp[0], xp[1], math.atan2(xp[3], xp[2])])

    def dot_plan(self, t):
        t = self._target_step(t)
        (ct, cx), (pt, px) = self._interval(t)
        xd = (cx - px) / ((ct - pt) * self.dt)
        return np.array([xd[0], xd[1], (xd[2] - xd[3]) / (xd[2] ** 2 + xd[3] ** 2)])
