#!/usr/bin/env python3
"""The reference's TorqueDistributionTask WITHOUT the simulator, assembled from the reference's own statements
(build container only; needs /root/reference). Used by make_golden_ref.py to write refexec_blocks.npz and
refexec_cosim_*.npz.

The env modules cannot be imported (omni / pxr / gym at module level) and the methods of the
step path mix pure torch arithmetic with calls into Isaac Sim. make_golden_ref.extract() takes whole functions; this
module goes one level down: it takes the methods of `TorqueDistributionTask` out of the syntax tree of a reference
env file and removes, by a mechanical rule, exactly the statements that cannot run here:

  R1  `self.world.is_playing()` is replaced by the constant True (the simulator is playing).
  R2  a simple statement is removed when it touches the simulator (`self._robots`, `self.world`, `self._env`,
      `SimulationContext`) or loads a global name that neither torch / math / numpy / builtins nor the reference's
      own pure helpers (`get_axis_params` KAMMA.py:911, `quat_rotate_inverse` my_utils/terrain_utils.py:581,
      `RLTask.cleanup` rl_task.py:145) provide: `quat_apply`, `quat_mul`, `torch_rand_float`, `normalize`
      (omni.isaac.core.utils.torch, not under /root/reference), `spaces` (gym), `RLTask.__init__`. print() calls are
      removed too.
  R3  a statement that reads a local whose every assignment was removed is removed as well — unless the harness
      hands that value in as a GIVEN (a module global of the compiled code): `projected_gravity` for is_done,
      `forward` / `up_dir` for the projection tail, `x_vel` for the uniform command draw, `points` (the scan lattice after
      `quat_apply_yaw`) for get_heights. Givens are data computed
      by the caller (here: by the oracle's quaternion helpers / the fixture's uniform draws), never code.
  R4  a compound statement whose header is removable goes as a whole; otherwise its body is filtered recursively.

Every other statement is compiled UNCHANGED and runs in its original order inside its original method, on a `self`
that is an instance of a class made of those methods. No stand-in is written for any absent function: what the
removed statements would have produced (the physics state after `SimulationContext.step`, the uniform draws of
`torch_rand_float`, the yaw quaternion of `quat_mul`) is written into the tensors by the caller between the calls,
the way the simulator's tensor views change under the reference's feet. `lines()` reports, per method, which source
lines run and which were removed; the fixtures store that list.
"""
import ast
import builtins
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference/src"
SIM_ATTRS = {"world", "_robots", "_env"}
METHODS = ["__init__", "update_config", "post_reset", "reset_idx", "sample_velocity_command", "refresh_dof_state_tensors",
           "refresh_body_state_tensors", "pre_physics_step", "post_physics_step", "push_robots", "is_done", "calculate_metrics",
           "get_observations", "get_heights"]
GIVENS = {"is_done": {"projected_gravity"}, "post_physics_step": {"forward", "up_dir"}, "sample_velocity_command": {"x_vel"},
          "get_heights": {"points"}}


class _Playing(ast.NodeTransformer):
    """R1: self.world.is_playing() -> True"""

    def visit_Call(self, node):
        self.generic_visit(node)
        f = node.func
        if (isinstance(f, ast.Attribute) and f.attr == "is_playing" and isinstance(f.value, ast.Attribute) and f.value.attr == "world"
                and isinstance(f.value.value, ast.Name) and f.value.value.id == "self"):
            return ast.copy_location(ast.Constant(True), node)
        return node


def _loads(node):
    bound = {a.arg for n in ast.walk(node) if isinstance(n, ast.arguments) for a in n.args + n.kwonlyargs}
    bound |= {n.id for n in ast.walk(node) if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del))}
    return {n.id for n in ast.walk(node) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)} - bound, bound


def _touches_sim(node):
    for n in ast.walk(node):
        if isinstance(n, ast.Attribute) and n.attr in SIM_ATTRS and isinstance(n.value, ast.Name) and n.value.id == "self":
            return True
        if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id == "print":
            return True
    return False


def _stores(node):
    return {n.id for n in ast.walk(node) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store)}


class _Filter:
    def __init__(self, known):
        self.known = known
        self.kept, self.dropped = [], []

    def bad(self, node):
        if _touches_sim(node):
            return True
        loads, _ = _loads(node)
        return bool(loads - self.known)

    def body(self, stmts):
        out = []
        for s in stmts:
            if isinstance(s, (ast.If, ast.While)):
                if self.bad(s.test):
                    self.dropped.append((s.lineno, s.end_lineno)); continue
                s.body = self.body(s.body) or [ast.Pass()]
                s.orelse = self.body(s.orelse)
                self.kept.append((s.lineno, s.lineno)); out.append(s)
            elif isinstance(s, ast.For):
                if self.bad(s.iter):
                    self.dropped.append((s.lineno, s.end_lineno)); continue
                self.known = self.known | _stores(s.target)
                s.body = self.body(s.body) or [ast.Pass()]
                s.orelse = self.body(s.orelse)
                self.kept.append((s.lineno, s.lineno)); out.append(s)
            elif isinstance(s, (ast.With, ast.Try)):
                raise NotImplementedError(f"line {s.lineno}: with / try in a step-path method")
            elif isinstance(s, ast.Expr) and isinstance(s.value, ast.Constant):
                continue  # docstring
            else:
                if self.bad(s):
                    self.dropped.append((s.lineno, s.end_lineno))
                else:
                    self.kept.append((s.lineno, s.end_lineno)); out.append(s)
        return out


def filter_method(fn, global_names, givens):
    """R2-R4 to a fixpoint: names assigned only by removed statements stop being known."""
    import copy
    params = {a.arg for a in fn.args.args + fn.args.kwonlyargs}
    base = set(dir(builtins)) | set(global_names) | params | set(givens)
    assigned = _stores(fn)
    while True:
        work = copy.deepcopy(fn)
        f = _Filter(base | assigned)
        work.body = f.body(work.body) or [ast.Pass()]
        now = _stores(work)
        if now == assigned:
            shared = sorted(set(givens) & now)
            if shared:  # a given that another branch of the method also assigns: one variable, the module global
                work.body.insert(0, ast.Global(names=shared))
            work.decorator_list = []
            work.returns = None
            for a in work.args.args:
                a.annotation = None
            return work, f.kept, f.dropped
        assigned = now


def _module_function(path, name):
    tree = ast.parse(open(path).read(), filename=path)
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    node.decorator_list = []
    return node


def _class_method(path, cls, name):
    tree = ast.parse(open(path).read(), filename=path)
    c = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    return next(n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == name)


class HarnessBase:
    """what omni's BaseTask / the wrapper give the task: attribute plumbing only, no arithmetic"""
    num_envs = property(lambda self: self._num_envs)
    device = property(lambda self: self._device)
    num_actions = property(lambda self: self._num_actions)
    num_observations = property(lambda self: self._num_observations)
    num_states = property(lambda self: 0)

    def get_states(self):  # RLTask API for asymmetric actor-critic; unused by this task
        return None


def build(variant_file):
    """-> (class, namespace, lines) for src/my_envs/<variant_file>.py"""
    path = os.path.join(REF, "my_envs", variant_file + ".py")
    tree = _Playing().visit(ast.parse(open(path).read(), filename=path))
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TorqueDistributionTask")
    ns = {"torch": torch, "math": math, "np": np, "HarnessBase": HarnessBase}
    helpers = [_module_function(path, "get_axis_params"), _module_function(os.path.join(REF, "my_utils", "terrain_utils.py"), "quat_rotate_inverse")]
    hm = ast.Module(body=helpers, type_ignores=[])
    ast.fix_missing_locations(hm)
    exec(compile(hm, path, "exec"), ns)
    body, lines = [], {}
    for fn in cls.body:
        if isinstance(fn, ast.FunctionDef) and fn.name in METHODS:
            work, kept, dropped = filter_method(fn, ns.keys(), GIVENS.get(fn.name, ()))
            body.append(work)
            lines[fn.name] = {"kept": kept, "dropped": dropped, "span": (fn.lineno, fn.end_lineno)}
    cleanup = _class_method(os.path.join(REF, "my_envs", "rl_task.py"), "RLTask", "cleanup")
    work, kept, dropped = filter_method(cleanup, ns.keys(), ())
    assert not dropped
    body.append(work)
    lines["RLTask.cleanup"] = {"kept": kept, "dropped": dropped, "span": (cleanup.lineno, cleanup.end_lineno)}
    new = ast.ClassDef(name="RefTask", bases=[ast.Name("HarnessBase", ast.Load())], keywords=[], body=body, decorator_list=[])
    mod = ast.Module(body=[new], type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, path, "exec"), ns)
    return ns["RefTask"], ns, lines


def lines_report(lines):
    out = []
    for m, d in lines.items():
        k = ",".join(f"{a}" if a == b else f"{a}-{b}" for a, b in sorted(d["kept"]))
        r = ",".join(f"{a}" if a == b else f"{a}-{b}" for a, b in sorted(d["dropped"]))
        out.append(f"{m} [{d['span'][0]}-{d['span'][1]}] runs: {k} | removed: {r or '-'}")
    return out


if __name__ == "__main__":
    for v in sys.argv[1:] or ["KAMMA"]:
        _, _, ln = build(v)
        print(v)
        for row in lines_report(ln):
            print("  ", row)
