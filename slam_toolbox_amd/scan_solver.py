"""Host-side mirror of the reference's solver plugin interface (karto::ScanSolver,
lib/karto_sdk/include/karto_sdk/Mapper.h:954-1066, implemented by solver_plugins::CeresSolver in
solvers/ceres_solver.cpp) over the C ABI of libkartohip.so.

Method names, argument meaning and error behaviour follow the reference: failures are reported (a
warning string is kept in `last_warning`) and the call returns leaving state unchanged, exactly like the
RCLCPP_WARN-and-return paths of ceres_solver.cpp:219-225, 249-254, 354-361, 420-424, 442-447."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def link_info(pose1, pose2, covariance):
    """LinkInfo::Update (Mapper.h:174-188) -> (pose difference, covariance in the frame of pose1)."""
    diff = np.zeros(3)
    cov = np.zeros(9)
    capi.check(capi.lib().kh_link_info(_d(pose1), _d(pose2), _d(covariance).reshape(9), diff, cov), "kh_link_info")
    return diff, cov.reshape(3, 3)


class HipSpaSolver:
    """solver_plugins::HipSpaSolver : karto::ScanSolver"""

    def __init__(self, device: int = 0, options: dict = None):
        h = C.c_void_p()
        capi.check(capi.lib().kh_spa_create(device, C.byref(h)), "kh_spa_create")
        self._h = h
        self.last_warning = ""
        self.summary = None
        if options:
            self.Configure(options)

    # ---- karto::ScanSolver -----------------------------------------------------------------
    def Configure(self, options: dict):
        """CeresSolver::Configure: the option values hard-wired at ceres_solver.cpp:157-186 can be overridden
        (e.g. the `tight` tolerances used by the parity tests)."""
        o = capi.KhSpaOptions()
        capi.lib().kh_spa_options_default(C.byref(o))
        for k, v in options.items():
            if not hasattr(o, k):
                raise KeyError(k)
            if k == "loss_function" and isinstance(v, str):
                # the `ceres_loss_function` strings of ceres_solver.cpp:82-94; anything else = squared loss
                v = {"HuberLoss": capi.KH_LOSS_HUBER, "CauchyLoss": capi.KH_LOSS_CAUCHY}.get(v, capi.KH_LOSS_NONE)
            setattr(o, k, v)
        capi.check(capi.lib().kh_spa_set_options(self._h, C.byref(o)), "kh_spa_set_options")

    def AddNode(self, unique_id: int, corrected_pose):
        capi.check(capi.lib().kh_spa_add_node(self._h, int(unique_id), _d(corrected_pose)), "kh_spa_add_node")

    def AddConstraint(self, source_id: int, target_id: int, pose_difference, covariance):
        rc = capi.lib().kh_spa_add_constraint(self._h, int(source_id), int(target_id), _d(pose_difference),
                                              _d(covariance).reshape(9))
        if rc == capi.KH_ERR_NOT_FOUND:
            self.last_warning = "CeresSolver: Failed to add constraint, could not find nodes."
            return
        capi.check(rc, "kh_spa_add_constraint")

    def RemoveNode(self, unique_id: int):
        rc = capi.lib().kh_spa_remove_node(self._h, int(unique_id))
        if rc == capi.KH_ERR_NOT_FOUND:
            self.last_warning = f"RemoveNode: Failed to find node matching id {unique_id}"
            return
        capi.check(rc, "kh_spa_remove_node")

    def RemoveConstraint(self, source_id: int, target_id: int):
        rc = capi.lib().kh_spa_remove_constraint(self._h, int(source_id), int(target_id))
        if rc == capi.KH_ERR_NOT_FOUND:
            self.last_warning = f"RemoveConstraint: Failed to find residual block for {source_id} {target_id}"
            return
        capi.check(rc, "kh_spa_remove_constraint")

    def set_debug(self, check_linear_solves: bool = False, factor_kernels: int = 0, gather_children: bool = False,
                  extend_add_pass: bool = False, phase_timing: bool = False):
        """kh_spa_set_debug: residual check of every linear solve; numeric kernels 0 default, 3 level pipeline, 2 panel pairs;
        level pipeline reading ALL the children's update matrices in place / summing them in with an extend-add launch per level
        (default: k_syrk adds a front's update matrix straight into its parent); HIP events around the phases of an iteration
        (the *_gpu_ms fields of the summary)"""
        flags = (1 if check_linear_solves else 0) | (int(factor_kernels) << 4) | (256 if gather_children else 0) | (512 if extend_add_pass else 0) | (2 if phase_timing else 0)
        capi.check(capi.lib().kh_spa_set_debug(self._h, flags), "kh_spa_set_debug")

    def Compute(self):
        s = capi.KhSpaSummary()
        rc = capi.lib().kh_spa_compute(self._h, C.byref(s))
        self.summary = {k: getattr(s, k) for k, _ in capi.KhSpaSummary._fields_}
        if rc == capi.KH_ERR_NOT_FOUND:
            self.last_warning = "CeresSolver: Ceres was called when there are no nodes. This shouldn't happen."
            return self.summary
        if rc == capi.KH_ERR_SOLVER:
            self.last_warning = "CeresSolver: Ceres could not find a usable solution to optimize."
            return self.summary
        capi.check(rc, "kh_spa_compute")
        return self.summary

    def iteration_log(self):
        """(n, 8) array: the trust-region iterations of the last Compute() (kh_spa_iteration_log: iteration, cost, candidate cost,
        model cost change, radius used, radius after, step norm, verdict)."""
        n = C.c_int32()
        capi.check(capi.lib().kh_spa_iteration_log(self._h, 0, None, C.byref(n)), "kh_spa_iteration_log")
        rows = np.zeros((max(1, n.value), 8))
        capi.check(capi.lib().kh_spa_iteration_log(self._h, n.value, rows.ctypes.data_as(C.c_void_p), C.byref(n)), "kh_spa_iteration_log")
        return rows[:n.value]

    def GetCorrections(self):
        """IdPoseVector: list of (unique id, pose)."""
        n = C.c_int32()
        capi.check(capi.lib().kh_spa_get_corrections(self._h, C.byref(n), None, None), "kh_spa_get_corrections")
        ids = np.zeros(max(1, n.value), dtype=np.int32)
        poses = np.zeros(max(1, n.value) * 3)
        capi.check(capi.lib().kh_spa_get_corrections(self._h, C.byref(n), ids.ctypes.data_as(C.c_void_p),
                                                     poses.ctypes.data_as(C.c_void_p)), "kh_spa_get_corrections")
        return [(int(ids[i]), poses[3 * i:3 * i + 3].copy()) for i in range(n.value)]

    def Clear(self):
        capi.check(capi.lib().kh_spa_clear(self._h), "kh_spa_clear")

    def Reset(self):
        capi.check(capi.lib().kh_spa_reset(self._h), "kh_spa_reset")

    def ModifyNode(self, unique_id: int, pose):
        capi.lib().kh_spa_modify_node(self._h, int(unique_id), _d(pose))

    def GetNodeOrientation(self, unique_id: int):
        p = np.zeros(3)
        rc = capi.lib().kh_spa_get_node(self._h, int(unique_id), p)
        return None if rc != capi.KH_OK else float(p[2])

    def getGraph(self):
        """unordered_map<int, Eigen::Vector3d>: id -> (x, y, yaw) of every node."""
        return {i: p for i, p in self._all_nodes()}

    # ---- multi-GPU -------------------------------------------------------------------------------
    def SetCommunicator(self, comm):
        """Edge-block sharded linearisation with the all-reduce INSIDE the library: ncclAllReduce of RCCL on `comm`
        (slam_toolbox_amd.comm.Communicator; None switches sharding off).  The communicator must outlive the solver's
        use of it."""
        self._comm = comm
        capi.check(capi.lib().kh_spa_set_comm(self._h, comm.handle if comm is not None else None), "kh_spa_set_comm")

    def enable_sharding(self, rank: int, world: int, group=None):
        """Edge-block sharded linearisation: every rank holds the same graph,
        rank r linearises edges [E r / world, E (r+1) / world) and the partial normal equations (H followed
        by g, one device buffer) are summed in place by torch.distributed.all_reduce -- RCCL over xGMI with
        backend "nccl", gloo in the single-GPU functional test.  Factorisation and LM control stay
        replicated, so all ranks return the same poses."""
        import torch
        import torch.distributed as dist

        class _DevView:                       # zero-copy view of the library's buffer as a torch tensor
            def __init__(self, ptr, count):
                self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}

        def hook(_user, ptr, count, _stream):
            try:
                # the library's launches are on its own stream: order the collective after them on the host,
                # run it on torch's stream, and hand the buffer back only when it is done
                torch.cuda.synchronize()
                t = torch.as_tensor(_DevView(int(ptr), int(count)), device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                torch.cuda.synchronize()
                return 0
            except Exception as exc:          # never raise through the C frame
                self.last_warning = f"all-reduce failed: {exc}"
                return 1
        self._hook = capi.ALLREDUCE_FN(hook) if world > 1 else capi.ALLREDUCE_FN(0)
        capi.check(capi.lib().kh_spa_set_sharding(self._h, int(rank), int(world), self._hook, None), "kh_spa_set_sharding")

    # ---- conveniences for tests / bench ---------------------------------------------------------
    def _all_nodes(self):
        out = []
        for i in self._ids:
            p = np.zeros(3)
            if capi.lib().kh_spa_get_node(self._h, int(i), p) == capi.KH_OK:
                out.append((int(i), p))
        return out

    _ids: list = []

    def load(self, poses, edges, z, cov):
        """SlamToolbox::loadSerializedPoseGraph (src/slam_toolbox_common.cpp:952-1017): Reset, AddNode for
        every vertex, AddConstraint for every edge."""
        self.Reset()
        poses = _d(poses)
        self._ids = list(range(poses.shape[0]))
        L = capi.lib()
        for i in range(poses.shape[0]):
            capi.check(L.kh_spa_add_node(self._h, i, poses[i]), "kh_spa_add_node")
        edges = np.asarray(edges)
        z = _d(z)
        cov = _d(cov).reshape(-1, 9)
        for e in range(edges.shape[0]):
            capi.check(L.kh_spa_add_constraint(self._h, int(edges[e, 0]), int(edges[e, 1]), z[e], cov[e]),
                       "kh_spa_add_constraint")

    def poses(self):
        return np.asarray([p for _, p in self._all_nodes()])

    # ---- pose-graph files (the reference's counterpart is the Boost archive read by
    # SlamToolbox::loadSerializedPoseGraph, src/slam_toolbox_common.cpp:952-1017) ------------------------
    def save_graph(self, path: str, binary: bool = False):
        capi.check(capi.lib().kh_spa_save(self._h, os.fsencode(path),
                                          capi.KH_GRAPH_BINARY if binary else capi.KH_GRAPH_TEXT), "kh_spa_save")

    def load_graph(self, path: str):
        """Reset, then AddNode / AddConstraint for every record of the file (text or binary, detected)."""
        capi.check(capi.lib().kh_spa_load(self._h, os.fsencode(path)), "kh_spa_load")
        self._ids = self.node_arrays()[0].tolist()

    def node_arrays(self):
        """(ids (n,), poses (n, 3)) of every node in AddNode order, one call."""
        n = capi.lib().kh_spa_num_nodes(self._h)
        ids, poses = np.zeros(max(n, 1), dtype=np.int32), np.zeros((max(n, 1), 3))
        capi.check(capi.lib().kh_spa_get_nodes(self._h, ids.ctypes.data_as(C.c_void_p), poses.ctypes.data_as(C.c_void_p)),
                   "kh_spa_get_nodes")
        return ids[:n], poses[:n]

    def nodes_in_order(self):
        out, L, i = [], capi.lib(), C.c_int32()
        for k in range(L.kh_spa_num_nodes(self._h)):
            p = np.zeros(3)
            capi.check(L.kh_spa_get_node_at(self._h, k, C.byref(i), p), "kh_spa_get_node_at")
            out.append((i.value, p))
        return out

    def constraints_in_order(self):
        """[(id_a, id_b, z (3), information upper triangle (6))] in AddConstraint order."""
        out, L, a, b = [], capi.lib(), C.c_int32(), C.c_int32()
        for k in range(L.kh_spa_num_constraints(self._h)):
            z, w = np.zeros(3), np.zeros(6)
            capi.check(L.kh_spa_get_constraint(self._h, k, C.byref(a), C.byref(b), z, w), "kh_spa_get_constraint")
            out.append((a.value, b.value, z, w))
        return out

    def AddConstraintInformation(self, source_id: int, target_id: int, pose_difference, information_upper):
        capi.check(capi.lib().kh_spa_add_constraint_information(self._h, int(source_id), int(target_id),
                                                                _d(pose_difference), _d(information_upper)),
                   "kh_spa_add_constraint_information")

    def close(self):
        if self._h:
            capi.lib().kh_spa_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
