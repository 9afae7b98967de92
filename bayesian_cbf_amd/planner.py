"""bayes_cbf/planner.py:19-64 `PiecewiseLinearPlanner`, same arithmetic, torch tensors on any device."""
import torch


class PiecewiseLinearPlanner:
    def __init__(self, x0, x_goal, numSteps, dt, frac_time_to_reach_goal=0.7):
        self.x0, self.x_goal, self.numSteps, self.dt = x0, x_goal, numSteps, dt
        assert numSteps >= 3
        self.frac_time_to_reach_goal = frac_time_to_reach_goal
        xdiff = x_goal[..., :2] - x0[..., :2]
        t2 = min(int(numSteps * frac_time_to_reach_goal), numSteps - 1)
        self._cps = [(t2, torch.cat([x_goal[..., :2], xdiff / xdiff.norm(dim=-1, keepdim=True)], dim=-1)),
                     (numSteps, torch.cat([x_goal[..., :2], x_goal[..., 2:].cos(), x_goal[..., 2:].sin()], dim=-1))]

    def _interval(self, t):
        prev_t = 0
        prev_x = torch.cat([self.x0[..., :2], self.x0[..., 2:].cos(), self.x0[..., 2:].sin()], dim=-1)
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
        return torch.cat([xp[..., :2], torch.atan2(xp[..., 3:4], xp[..., 2:3])], dim=-1)

    def dot_plan(self, t):
        t = self._target_step(t)
        (ct, cx), (pt, px) = self._interval(t)
        xd = (cx - px) / ((ct - pt) * self.dt)
        return torch.cat([xd[..., :2], (xd[..., 2:3] - xd[..., 3:4]) / (xd[..., 2:4] ** 2).sum(-1, keepdim=True)], dim=-1)
