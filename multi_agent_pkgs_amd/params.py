"""ctypes mirror of `hdsm_params` (include/hdsm.h) and the reference's shipped configurations.

The field values of :func:`agile_params` are the ROS parameters of
``multi_agent_planner/config/agent_agile_config.yaml`` of the reference, turned into solver bounds the way
``Agent::InitializePlannerParameters`` does (agent_class.cpp:2169-2188).
"""
import ctypes as C

HDSM_MAX_HOR = 16
HDSM_MAX_POLY = 8
HDSM_MAX_ROWS_STATIC = 32
HDSM_INF = 1e100

HDSM_OPTIMAL, HDSM_LIMIT, HDSM_NO_SOLUTION = 0, 1, 2


class HdsmParams(C.Structure):
    _fields_ = [
        ("n_hor", C.c_int32),
        ("poly_hor", C.c_int32),
        ("rk4", C.c_int32),
        ("max_rows_static", C.c_int32),
        ("dt", C.c_double),
        ("drag", C.c_double * 3),
        ("r_u", C.c_double),
        ("r_x", C.c_double * 9),
        ("r_n", C.c_double * 9),
        ("x_lb", C.c_double * 9),
        ("x_ub", C.c_double * 9),
        ("u_lb", C.c_double * 3),
        ("u_ub", C.c_double * 3),
        ("drone_radius", C.c_double),
        ("drone_z_offset", C.c_double),
        ("plane_perturb", C.c_double),
        ("max_nodes", C.c_int32),
        ("max_qp_iters", C.c_int32),
        ("feas_tol_fixed", C.c_double),
        ("solver_tol", C.c_double),
        ("warm_start", C.c_int32),
        ("threads_per_instance", C.c_int32),
        ("prefilter_min_agents", C.c_int32),
        ("duo_min_instances", C.c_int32),
        ("presweep", C.c_int32),
        ("branch_rule", C.c_int32),
        ("launch_order", C.c_int32),
        ("stage_radius", C.c_double),
        ("time_limit_s", C.c_double),
        ("mip_gap", C.c_double),
    ]

    def copy(self):
        other = HdsmParams()
        C.memmove(C.byref(other), C.byref(self), C.sizeof(self))
        return other


def make_params(n_hor=10, poly_hor=4, rk4=False, dt=0.1, drag=(0.0, 0.0, 0.0), r_u=0.01,
                r_x=(100.0, 100.0, 100.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0),
                r_n=(100.0, 100.0, 100.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0),
                max_vel=20.0, max_acc_xy=15.0, min_acc_xy=-15.0, max_acc_z=15.0, min_acc_z=-15.0,
                max_jerk=60.0, drone_radius=0.25, drone_z_offset=0.25, plane_perturb=0.1,
                max_rows_static=18, max_nodes=0, max_qp_iters=0, feas_tol_fixed=1e-6, solver_tol=1e-9,
                warm_start=True, threads_per_instance=0, prefilter_min_agents=0, duo_min_instances=0, presweep=0,
                branch_rule=0, launch_order=0, stage_radius=0.0, time_limit_s=0.0, mip_gap=0.0):
    """Build an HdsmParams the way InitializePlannerParameters builds x_lb_/x_ub_/u_lb_/u_ub_ (n_x = 9)."""
    p = HdsmParams()
    p.n_hor, p.poly_hor, p.rk4, p.max_rows_static = n_hor, poly_hor, int(bool(rk4)), max_rows_static
    p.dt, p.r_u = dt, r_u
    for k in range(3):
        p.drag[k] = drag[k]
    for k in range(9):
        p.r_x[k], p.r_n[k] = r_x[k], r_n[k]
    lb = [-HDSM_INF] * 3 + [-max_vel] * 3 + [min_acc_xy, min_acc_xy, min_acc_z]
    ub = [HDSM_INF] * 3 + [max_vel] * 3 + [max_acc_xy, max_acc_xy, max_acc_z]
    for k in range(9):
        p.x_lb[k], p.x_ub[k] = lb[k], ub[k]
    for k in range(3):
        p.u_lb[k], p.u_ub[k] = -max_jerk, max_jerk
    p.drone_radius, p.drone_z_offset, p.plane_perturb = drone_radius, drone_z_offset, plane_perturb
    p.max_nodes, p.max_qp_iters = max_nodes, max_qp_iters
    p.feas_tol_fixed, p.solver_tol = feas_tol_fixed, solver_tol
    p.warm_start = int(bool(warm_start))
    p.threads_per_instance, p.prefilter_min_agents, p.duo_min_instances = threads_per_instance, prefilter_min_agents, duo_min_instances
    p.presweep, p.branch_rule, p.stage_radius, p.time_limit_s = presweep, branch_rule, stage_radius, time_limit_s
    p.launch_order, p.mip_gap = launch_order, mip_gap
    return p


def agile_params(n_hor=10, **over):
    """agent_agile_config.yaml (poly_hor 4, dt 0.1, jerk control) with n_hor overridden (BASELINE: 10 / 15)."""
    return make_params(n_hor=n_hor, **over)


def default_params(n_hor=9, **over):
    """agent_default_config.yaml: poly_hor 3, max_vel 9.5, acc +-20, jerk 30, drone radius 0.125."""
    kw = dict(poly_hor=3, max_vel=9.5, max_acc_xy=20.0, min_acc_xy=-20.0, max_acc_z=20.0, min_acc_z=-20.0,
              max_jerk=30.0, drone_radius=0.125, drone_z_offset=0.125)
    kw.update(over)
    return make_params(n_hor=n_hor, **kw)


class RefConfig(C.Structure):
    """ctypes mirror of hdsm_ref_config (reference-trajectory parameters, agent_class.cpp:2190-2248)."""
    _fields_ = [("path_vel_min", C.c_double), ("path_vel_max", C.c_double), ("sens_dist", C.c_double),
                ("sens_pot", C.c_double), ("sens_other_agents", C.c_double), ("path_vel_dec", C.c_double)]


def agile_ref_config(**over):
    """agent_agile_config.yaml: path_vel 4.5..9.0, sens_dist 0.05, sens_pot 0.18; sens_other_agents default 1.0."""
    kw = dict(path_vel_min=4.5, path_vel_max=9.0, sens_dist=0.05, sens_pot=0.18, sens_other_agents=1.0, path_vel_dec=0.0)
    kw.update(over)
    c = RefConfig()
    for k, v in kw.items():
        setattr(c, k, v)
    return c


class MapConfig(C.Structure):
    """hdsm_map_config (include/hdsm.h): the map pre-processing parameters of map_builder_default_config.yaml:8-10."""
    _fields_ = [("voxel_size", C.c_double), ("inflation_dist", C.c_double), ("potential_dist", C.c_double),
                ("potential_pow", C.c_int32), ("reserved0", C.c_int32)]


def default_map_config(voxel_size=0.3, inflation_dist=0.3, potential_dist=1.5, potential_pow=4):
    return MapConfig(voxel_size, inflation_dist, potential_dist, potential_pow, 0)
