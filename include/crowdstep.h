/*
 * crowdstep.h -- C ABI of libcrowdstep.so, the MI355X (gfx950) batched crowd-stepper.
 *
 * The reference (TommasoVandermeer/Social-Navigation-PyEnvs) has no FFI: its seam for this path is
 * the pure-array Python function  update_humans_parallel()  (social_gym/src/forces_parallel.py:185,
 * called from exactly one site, social_gym/src/motion_model_manager.py:360) plus the rvo2
 * PyRVOSimulator object API for ORCA (motion_model_manager.py:237-246, 386-394).  Every entry point
 * below names the reference interface it replaces.  All of them are batched over W independent
 * worlds; W = 1 with the reference's [N,13] arrays is the drop-in case.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.  Pointers named d_* are DEVICE
 *     pointers (hipMalloc / cs_malloc / torch.Tensor.data_ptr()); h_* are host pointers.
 *   - return 0 (CS_OK) or a negative cs_status; cs_last_error() gives the message (thread-local).
 *     Nothing throws or aborts across the ABI.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *     asynchronous on that stream unless stated otherwise.
 *   - arithmetic type: float32 ("f32").  Array layouts are the reference's row layouts:
 *       state row  S[13]  = px,py,theta,vx,vy,bvx,bvy,omega,r,m,gx,gy,vd       (agent.py:256-258)
 *       param row  P[20]  = relax_t,Ai,Aw,Bi,Bw,Ci,Cw,Di,Dw,Ei,k1,k2,lambda,gamma,ns,ns1,ko,kd,
 *                           alpha,k_lambda                                      (agent.py:268-388)
 *       goals [n][G][2] NaN padded; obstacles [O][Smax][2][2] NaN padded
 *   - a world has n humans (+1 trailing robot row when `robot_row` is set) = `rows` rows.
 */
#ifndef CROWDSTEP_H
#define CROWDSTEP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cs_status {
    CS_OK = 0,
    CS_ERR_ARG = -1,      /* bad shape / null pointer / unsupported size                       */
    CS_ERR_TYPE = -2,     /* model type outside 0..8 (the reference raises ValueError, :211)    */
    CS_ERR_HIP = -3,      /* HIP runtime error, message in cs_last_error()                      */
    CS_ERR_NO_DEVICE = -4 /* no gfx950 device visible                                           */
} cs_status;

/* model ids = index into the reference's SFMS list (motion_model_manager.py:15-17) */
enum {
    CS_SFM_HELBING = 0, CS_SFM_GUO = 1, CS_SFM_MOUSSAID = 2,
    CS_HSFM_FARINA = 3, CS_HSFM_GUO = 4, CS_HSFM_MOUSSAID = 5,
    CS_HSFM_NEW = 6, CS_HSFM_NEW_GUO = 7, CS_HSFM_NEW_MOUSSAID = 8,
    CS_ORCA = 9, /* HUMAN_MODELS[9] (social_nav_gym.py:11-12); only cs_step / cs_peek accept it */
    CS_SOCIAL_MOMENTUM = 10 /* MotionModelManager("social_momentum") (motion_model_manager.py:247-251, 395-404;
                               social_gym/src/social_momentum.py); only cs_step / cs_peek accept it */
};

/* memory layout of a state array: element (world w, row a, field f) lives at
 *   base[(w*rows + a) * agent_stride + f * field_stride]
 * CS_LAYOUT_AOS: the reference's [W][rows][13]   (agent_stride 13, field_stride 1)
 * CS_LAYOUT_SOA: 13 planes [13][W*rows]          (agent_stride 1,  field_stride W*rows) */
enum { CS_LAYOUT_AOS = 0, CS_LAYOUT_SOA = 1 };

/* flag bits of cs_worlds.flags */
enum {
    CS_ALL_PARAMS_EQUAL = 1 << 0, /* update_humans_parallel(all_params_equal=True): pair forces use P[0]  */
    CS_ROBOT_ROW        = 1 << 1, /* last_is_robot=True: last row is the robot, never updated by the model */
    CS_PARAMS_SHARED    = 1 << 2, /* d_params is one [n][20] block shared by all worlds                    */
    CS_OBSTACLES_SHARED = 1 << 3, /* d_obstacles is one [O][Smax][2][2] block shared by all worlds         */
    CS_RESPAWN          = 1 << 4, /* parallel-traffic respawn after every substep (mmm.py:407-422)         */
    CS_ROBOT_UNICYCLE   = 1 << 5  /* d_action rows are (v, r) instead of (vx, vy) (robot_agent.py:119-136) */
};

/* Descriptor of W resident worlds; every pointer is a caller-owned device buffer. */
typedef struct cs_worlds {
    int32_t W;          /* worlds                                                                */
    int32_t n;          /* humans per world                                                      */
    int32_t G;          /* goal slots per human                                                  */
    int32_t O;          /* polygons (0 = no walls)                                               */
    int32_t Smax;       /* segment slots per polygon                                             */
    int32_t type;       /* 0..8                                                                  */
    int32_t flags;      /* CS_* bits                                                             */
    int32_t layout;     /* CS_LAYOUT_*  of d_state                                               */
    float*  d_state;    /* [W][rows][13] (or SoA planes)  in/out                                 */
    float*  d_goals;    /* [W][n][G][2]                   in/out (rotated on goal switch)        */
    const float* d_params;    /* [W][n][20] or [n][20]                                           */
    const float* d_safety;    /* [W][rows]                                                       */
    const float* d_obstacles; /* [W][O][Smax][2][2] or shared, NULL when O == 0                  */
    float*  d_robot;    /* [W][13] robot safe-state rows or NULL.  With CS_ROBOT_ROW it is the
                           source of the last row before every substep (mmm.py:359)             */
    const int32_t* d_world_flags; /* optional [W]: bit0 = respawn enabled in this world (NULL: the
                           CS_RESPAWN bit applies to every world; hybrid batches mix both kinds)   */
    float   respawn_bound_x, respawn_bound_y; /* respawn_bounds (social_nav_sim.py:360)           */
    /* ORCA only (type == CS_ORCA): ORCA_DEFAULTS of motion_model_manager.py:14 */
    float   orca_neighbor_dist, orca_time_horizon, orca_time_horizon_obst;
    int32_t orca_max_neighbors;
    /* social momentum only (type == CS_SOCIAL_MOMENTUM): n_actions of motion_model_manager.py:249 (0 = 20) */
    int32_t sm_n_actions;
    /* ORCA only: static obstacles as RVO2 keeps them after addObstacle / processObstacles (motion_model_manager.py:244-246):
     * [orca_n_vertices][8] records  point.x, point.y, unitDir.x, unitDir.y, isConvex, next, prev, 0  (polygon vertices in
     * the order given, counter-clockwise; next / prev = vertex indices), shared by all worlds.  NULL / 0 = no obstacles. */
    const float* d_orca_vertices;
    int32_t orca_n_vertices;
    /* ORCA only: RVO2 keeps neighborDist, maxNeighbors, timeHorizon, timeHorizonObst PER AGENT (RVOSimulator::addAgent(position,
     * neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, radius, maxSpeed, velocity): the call at motion_model_manager.py:241,
     * where the reference passes ORCA_DEFAULTS for everyone).  [W][rows][4] floats in that order, or NULL = the four scalars above for
     * every agent.  With it, orca_max_neighbors must be the LARGEST maxNeighbors and orca_neighbor_dist the largest neighborDist. */
    const float* d_orca_agent_params;
    /* ORCA only (ABI 4): the arithmetic of the register-resident build for THESE worlds -- CS_ORCA_MATH_* below.  0 (a zeroed struct) = the
     * process default: CROWDSTEP_ORCA_MATH=exact|fast|fma if set, else exact.  A field of the context, not a process-wide switch: two
     * environments of one process may differ, and the mode is part of what a captured HIP graph froze. */
    int32_t orca_math;
} cs_worlds;

/* cs_worlds.orca_math: how the register-resident ORCA build (k_orca_step<FAST10>: maxNeighbors = 10, no static obstacles, one RVO2
 * parameter set -- what motion_model_manager.py:14, 237-246 always creates) divides and takes square roots in RVO2's linearProgram1 / 2 / 3
 * and line construction (reached through rvo2.PyRVOSimulator.doStep, motion_model_manager.py:387; float32 like RVO2):
 *   EXACT  correctly rounded divide / sqrt, no FMA contraction: bit-identical to oracle/orca_oracle.c.  THE DEFAULT.
 *   FAST   v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 (1 ulp each);
 *   FMA    fast, and determinants / dot products / point + t * direction as mul + fma.
 * FAST / FMA are opt-in: they stay within north_star's 1e-5 per step except on the agent-substeps float32 does not determine (a decision
 * edge of the linear programme or an ill-conditioned intersection; tests/test_gpu_orca_fast.py classifies every one).  The generic ORCA
 * builds (other maxNeighbors, static obstacles, per-agent parameters, the grid path, the robot's own ORCA model) are always exact. */
enum { CS_ORCA_MATH_DEFAULT = 0, CS_ORCA_MATH_EXACT = 1, CS_ORCA_MATH_FAST = 2, CS_ORCA_MATH_FMA = 3 };

/* ---------------------------------------------------------------- runtime / memory helpers */
const char* cs_last_error(void);
/* The ABI this header describes.  A host binding must refuse a library whose cs_abi_version() differs: struct layouts (cs_worlds,
 * cs_gym_book, cs_stage_book) and argument lists change between versions (social_navigation_pyenvs_amd/_lib.py load() does). */
#define CS_ABI_VERSION 4
int cs_abi_version(void);
int cs_device_count(int* count);
int cs_set_device(int device);
int cs_device_name(int device, char* buf, size_t buflen);
/* PCI address "domain:bus:device.function" of a device: bench.py reports it per rank, so that a multi-GPU line proves which
 * card every rank sat on (one process per GPU, no collective on the data path) */
int cs_device_pci_bus_id(int device, char* buf, size_t buflen);
int cs_malloc(void** d_ptr, size_t bytes);
int cs_free(void* d_ptr);
int cs_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream);
int cs_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream);
int cs_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, void* stream);
int cs_memset(void* d_ptr, int value, size_t bytes, void* stream);
int cs_stream_create(void** stream);
/* priority: 0 = default, < 0 = the device's highest, > 0 = its lowest (background work beside a step stream: the refill passes of
 * cs_refill_staged_worlds); a stream of another priority also lands on another hardware queue */
int cs_stream_create_with_priority(void** stream, int priority);
int cs_stream_destroy(void* stream);
int cs_stream_sync(void* stream);
/* HIP events on the launch stream (bench.py times kernels with these) */
int cs_event_create(void** event);
int cs_event_destroy(void* event);
int cs_event_record(void* event, void* stream);
int cs_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */
int cs_event_query(void* event, int* done);                  /* *done = 1 when the work recorded before `event` has finished; never blocks */
int cs_stream_wait_event(void* stream, void* event);         /* later work of `stream` waits for `event` (no host sync) */

/* HIP graphs: capture the launches issued on `stream` between begin and end (e.g. K cs_step calls of a rollout) and
 * replay them with one launch -- removes the per-launch host gap of a launch-bound loop. */
int cs_graph_begin_capture(void* stream);
int cs_graph_end_capture(void* stream, void** graph_exec);
int cs_graph_launch(void* graph_exec, void* stream);
int cs_graph_destroy(void* graph_exec);

/* ---------------------------------------------------------------- the hot path
 *
 * cs_update_humans_parallel  replaces  update_humans_parallel(type, agents_state, goals, obstacles,
 *   agents_params, dt, safety_space, all_params_equal, last_is_robot) -> updated_state
 *   (forces_parallel.py:185-284), for W worlds at once, one Euler substep.
 *   d_state is read and mutated exactly like `agents_state` (goal columns on a goal switch and, for
 *   headed types, columns 3:5 = R(theta)*bv); d_goals is rotated in place; d_out receives the
 *   returned copy (robot row copied through).  d_out may equal w->d_state (in-place update, what
 *   `self.states = update_humans_parallel(...)` amounts to, motion_model_manager.py:360).
 *   w->d_robot is ignored here (the last row of d_state IS the robot row).
 */
int cs_update_humans_parallel(const cs_worlds* w, float dt, float* d_out, void* stream);

/*
 * cs_step  replaces the substep loop of SocialNavGym.step (social_nav_gym.py:240-245) /
 *   MotionModelManager.update_humans (motion_model_manager.py:354-367, 407-422), fused in ONE
 *   launch:  n_substeps x { robot.step(action, dt) ; states[-1] = robot row ; update_humans ;
 *   respawn }.  State is updated in place.
 *   d_action: [W][2] robot action held for the whole block, or NULL (robot does not move).
 */
int cs_step(const cs_worlds* w, float dt, int n_substeps, const float* d_action, void* stream);

/*
 * cs_step_trace  cs_step (same kernel build, same arithmetic, same in-place update -- the launch differs only in one non-NULL
 *   kernel argument) that also records every human's row AFTER every fused substep, so that a test can check each substep of a
 *   fused launch on its own against the reference's single-substep function (forces_parallel.py:185-284 + the respawn rule,
 *   motion_model_manager.py:407-422) restarted from the previous record.
 *   d_trace [n_substeps][W][rows][12] = px, py, theta, vx, vy, bvx, bvy, omega, gx, gy (state columns 10:12), goals[0].x, goals[0].y
 *   (rows = n + 1 with CS_ROBOT_ROW: the robot's record k is the robot as substep k + 1 sees it -- it moves before the humans,
 *   social_nav_gym.py:240-243 -- and as it stands at the end for the last record).
 *   Types 0..8, any world size (worlds beyond one block: the grid path records from HBM after every substep).
 */
int cs_step_trace(const cs_worlds* w, float dt, int n_substeps, const float* d_action, float* d_trace, void* stream);

/*
 * cs_peek  replaces MotionModelManager.get_next_human_observable_states(dt, theta_and_omega_visible)
 *   (motion_model_manager.py:691-709): one Euler step of size dt WITHOUT committing it.
 *   d_next: [W][n][8] rows  x, y, yaw, Vx, Vy, Omega, Gx, Gy  (get_human_states(include_goal=True,
 *   headed=False), :294-298); the caller slices [0,1,3,4] for the 4-column form.
 */
int cs_peek(const cs_worlds* w, float dt, float* d_next, void* stream);

/*
 * cs_collision_reward  replaces SocialNavSim.collision_detection_and_reaching_goal
 *   (social_nav_sim.py:949-984) + compute_reward_and_infos (:986-1029) for W worlds.
 *   Reads humans from w->d_state and the robot from w->d_robot (position, radius, goal).
 *   d_action [W][2] holonomic (vx, vy).  d_global_time [W].
 *   reward_cfg = {time_limit, success_reward, collision_penalty, discomfort_dist,
 *                 discomfort_penalty_factor}
 *   d_out [W][7] = collision, dmin, reaching_goal, reward, terminated, truncated, info_code
 *   (info_code: 0 Nothing, 1 Danger, 2 ReachGoal, 3 Collision, 4 Timeout; social_gym/src/info.py)
 */
int cs_collision_reward(const cs_worlds* w, const float* d_action, float T, const float* d_global_time,
                        const float* reward_cfg /* host, 5 floats */, float* d_out, void* stream);

/*
 * cs_lookahead  replaces compute_rotated_states_and_reward(action_space, next_humans_state, current_humans_state,
 *   current_robot_state, dt, theta_and_omega_visible) -> (rotated_states, rewards)
 *   (crowd_nav/policy/cadrl.py:42-83, with transform_state_to_agent_centric :13-39) for W robots at once.
 *   d_actions [A][2]; d_next [W][n][4] (px,py,vx,vy) or [W][n][6] (x,y,yaw,Vx,Vy,Omega) when theta_and_omega_visible;
 *   d_current [W][n][5] (px,py,vx,vy,r) or [W][n][7] (+theta,omega); d_robot [W][robot_stride] rows starting with
 *   px,py,vx,vy,r,gx,gy,v_pref.  Outputs d_rotated [W][A][n][13|15] (the value-network input), d_rewards [W][A].
 */
int cs_lookahead(int W, int n, int A, int theta_and_omega_visible, const float* d_actions, const float* d_next,
                 const float* d_current, const float* d_robot, int robot_stride, float dt, float* d_rotated,
                 float* d_rewards, void* stream);

/*
 * cs_generate_worlds  replaces, for W worlds at once and on the device, what SocialNavGym.reset does per world on the
 *   host (social_nav_gym.py:135-197):  np.random.seed(seed) ; [hybrid: np.random.choice + re-seed] ;
 *   generate_circular_crossing_setting / generate_parallel_traffic_scenario /
 *   generate_circular_crossing_with_static_obstacles (social_nav_sim.py:200-431) ; the HumanAgent rows reset_sim
 *   builds (:97-198) ; robot.set(...) (social_nav_gym.py:211-213).
 *   One lane per world restates numpy's legacy MT19937 stream (init_genrand seeding, 53-bit random_sample,
 *   uniform, choice-of-two) and the generators' rejection loops in f64, draw for draw; rows are rounded to f32 when
 *   stored.  d_seeds [W] = offset[phase] + case (social_nav_gym.py:135-137).  Writes w->d_state (humans, and the
 *   robot row with CS_ROBOT_ROW), w->d_goals ([W][n][G][2], NaN-padded), w->d_robot (if not NULL), w->d_world_flags
 *   (if not NULL: 1 for parallel-traffic worlds = respawn rule on).
 *   d_mask     optional [W]: only worlds with a non-zero entry are regenerated (auto-reset of finished episodes).
 *   d_status   optional [W]: 0 ok, 1 = a human could not be placed within max_tries (the reference loops forever),
 *              2 = parallel traffic too dense (ValueError in the reference, :318-319); such worlds are left untouched.
 *   d_scenario optional [W]: the scenario each world got (the hybrid choice).
 *   d_scratch  cs_generate_scratch_bytes(W) bytes (624 MT19937 words per world).
 */
enum {
    CS_SCN_CIRCULAR_CROSSING = 0,                  /* 'circle_crossing'                          */
    CS_SCN_PARALLEL_TRAFFIC = 1,                   /* 'parallel_traffic'                         */
    CS_SCN_CIRCULAR_CROSSING_STATIC_OBSTACLES = 2, /* 'circular_crossing_with_static_obstacles'  */
    CS_SCN_HYBRID = 3                              /* 'hybrid_scenario': choice of the first two */
};
typedef struct cs_generator {
    int32_t scenario;             /* CS_SCN_*                                                               */
    int32_t n;                    /* n_actors                                                               */
    int32_t insert_robot;         /* robot start / goal take part in the rejection tests (:284-287)         */
    int32_t randomize_attributes; /* uniform(0.5,1.5) speed and uniform(0.3,0.5) radius per human (:217-220) */
    int32_t randomize_positions;  /* circular crossing only: 0 = evenly spaced (:231-251)                    */
    int32_t max_tries;            /* bound of every rejection loop                                           */
    double  circle_radius, traffic_length, traffic_height, robot_radius;
    double  human_mass;           /* 75 (reset_sim, :167)                                                    */
    double  robot_mass, robot_desired_speed; /* RobotAgent defaults (agent.py)                               */
} cs_generator;
/* (an ORCA batch -- w->type == CS_ORCA -- also gets RVO2's preferred velocity in columns 5:7 of every generated row: update_goals_orca,
 * motion_model_manager.py:125-133, as the reference sets it when it builds the simulator) */
size_t cs_generate_scratch_bytes(int W);
int cs_generate_worlds(const cs_generator* gen, const cs_worlds* w, const uint32_t* d_seeds, const int32_t* d_mask,
                       int32_t* d_status, int32_t* d_scenario, void* d_scratch, void* stream);

/*
 * cs_laser_scan  replaces LaserSensor.get_laser_measurements(humans, walls) without the noise term
 *   (social_gym/src/sensors.py:51-66; disc hit :24-33, one-sided segment hit :35-49) for W sensors at once.
 *   Humans (x, y, radius) are read from w->d_state, walls from w->d_obstacles.  Sensor poses (x, y, yaw in columns
 *   0..2 of rows of `pose_stride` floats) come from d_pose, or from w->d_robot (robot safe-state rows) when d_pose
 *   is NULL.  Ray k of `samples` points at linspace(yaw - range/2, yaw + range/2, samples)[k] (:53).
 *   d_out [W][samples] = min(hit distance, max_distance); RobotAgent.get_laser_readings subtracts the robot radius
 *   afterwards (robot_agent.py:77-82) -- the host mirror does that.
 *   Errors: max_distance > 10 -> CS_ERR_ARG with the reference's message (sensors.py:13).
 */
int cs_laser_scan(const cs_worlds* w, const float* d_pose, int pose_stride, float range, int samples,
                  float max_distance, float* d_out, void* stream);

/*
 * cs_robot_model_step  replaces MotionModelManager.update_robot(t, dt) (motion_model_manager.py:615-653) for a robot that
 *   follows a HUMAN motion model, as set by set_robot_motion_model / SocialNavGym.set_human_motion_model_as_robot_policy
 *   (motion_model_manager.py:552-589, social_nav_sim.py:862-873): ONE Euler substep of the robot of every world, the robot half
 *   of the substep loop of SocialNavGym.imitation_learning_step (social_nav_gym.py:259-263; the human half is cs_step with
 *   n_substeps = 1 and no action, which picks the moved robot up from w->d_robot when it is visible).
 *   robot_type 0..8: compute_robot_forces (:591-613) with the single-agent force functions of forces.py and the Euler update
 *     (:72-86); robot_params = the robot's 20 parameters (agent.py:269 slots; host pointer).  d_robot_memory [W][2] keeps
 *     robot.desired_force between substeps (the reference leaves it untouched within one radius of the goal, forces.py:12-16);
 *     zero it when the robot is created.  Walls: w->d_obstacles.
 *   robot_type CS_ORCA: one doStep of the robot's own RVO2 simulator (:580-589, :641-653: humans with preferred velocity 0,
 *     the robot last) = the robot's ORCA solve against the humans and w->d_orca_vertices, with w->orca_* parameters;
 *     robot_params and d_robot_memory are ignored.  PARITY UNPINNED like cs_step with CS_ORCA.
 *   robot_margin: robot.safety_space (SFM: 0 or 0.01 + safety_space; ORCA: 0.01 + safety_space, :147-170, :588).
 *   d_human_margin [W][rows]: the humans' safety_space as the ROBOT's model sees it (SFM robot: human.safety_space;
 *     ORCA robot: 0.01 + safety_space); NULL = w->d_safety.
 *   Reads the humans from w->d_state; updates w->d_robot rows (x, y, yaw, Vx, Vy, BVx, BVy, Omega) and, with CS_ROBOT_ROW,
 *   the last state row.  Errors: robot_type outside 0..9 -> CS_ERR_TYPE (the reference raises, :589).
 */
int cs_robot_model_step(const cs_worlds* w, int32_t robot_type, const float* robot_params, float robot_margin,
                        const float* d_human_margin, float* d_robot_memory, float dt, void* stream);

/*
 * cs_robot_model_velocities  replaces MotionModelManager.update_robot(t, dt, just_velocities=True) (motion_model_manager.py:615-629
 *   with euler_*_single_agent_update(..., just_velocities=True), :72-85): the robot's velocities (linear or body + angular) are
 *   integrated by its SFM / HSFM model, its position and yaw stay (SocialNavSim moves the pose at its own rate with
 *   update_robot_pose, :655-659).  Arguments as cs_robot_model_step; robot_type 0..8, or CS_ORCA: the robot's doStep gives the new
 *   velocity and the robot's simulator agent is put back on robot.position (:641-653).
 */
int cs_robot_model_velocities(const cs_worlds* w, int32_t robot_type, const float* robot_params, float robot_margin,
                              const float* d_human_margin, float* d_robot_memory, float dt, void* stream);

/*
 * cs_imitation_block  replaces the substep loop of SocialNavGym.imitation_learning_step (social_nav_gym.py:259-263):
 *   n_substeps x { motion_model_manager.update_robot(t, dt) ; motion_model_manager.update_humans(t, dt) }, arguments as
 *   cs_robot_model_step.  With an invisible robot (no CS_ROBOT_ROW: the crowd does not see it) and SFM / HSFM models on both sides
 *   this is TWO launches: the crowd's n fused substeps, which also record what the robot's integrator sees of the humans at the
 *   start of every substep (positions and linear velocities, library-owned scratch: see cs_reserve_scratch), and the robot's n substeps against those
 *   records -- bit-identical to the alternating launches.  Otherwise (visible robot, ORCA on either side) the 2 n launches are
 *   issued in the reference's order.
 */
int cs_imitation_block(const cs_worlds* w, int32_t robot_type, const float* robot_params, float robot_margin,
                       const float* d_human_margin, float* d_robot_memory, float dt, int n_substeps, void* stream);

/*
 * Library-owned scratch.  Three entry points keep device scratch of their own: cs_imitation_block's fused form (what the robot sees
 * of the crowd, [n_substeps][W][n] x 16 B) and cs_step / cs_update_humans_parallel for worlds beyond one block (double-buffered
 * rows + the neighbour grid).  The block belongs to (device, stream, use): batches driven on different streams never share one.
 * Nothing is allocated, grown or freed while `stream` is capturing: an entry point that would have to returns CS_ERR_ARG
 * -- call cs_reserve_scratch(w, n_substeps, stream) (or the entry point itself once) before cs_graph_begin_capture.  A block
 * handed out during a capture is never freed behind a graph's back (growing it later retires the old block); cs_release_scratch
 * synchronises EVERY device that holds a block and frees them all -- only when no captured graph that used them will be replayed again.
 * Threads: one host thread per (device, stream) at a time; different streams never share a block.
 */
int cs_reserve_scratch(const cs_worlds* w, int n_substeps, void* stream);
int cs_release_scratch(void);

/*
 * cs_actual_collision_reward  replaces SocialNavGym.check_actual_collisions_and_goal (social_nav_gym.py:107-118) +
 *   compute_reward_and_infos (social_nav_sim.py:986-1029) for W worlds: distances of the CURRENT state (no swept test;
 *   collision when the smallest distance is <= 0).  Same reward_cfg and d_out layout as cs_collision_reward.
 */
int cs_actual_collision_reward(const cs_worlds* w, float T, const float* d_global_time, const float* reward_cfg /* host, 5 floats */,
                               float* d_out, void* stream);

/*
 * cs_update_humans_rk45  replaces MotionModelManager(runge_kutta=True).update_humans(t, dt) (motion_model_manager.py:374-384):
 *   scipy.integrate.solve_ivp(f_rk45_headed | f_rk45_not_headed, (t, t + dt), y0, method='RK45') of every world, i.e. the
 *   Dormand-Prince 5(4) pair with scipy's step-size control (rtol 1e-3, atol 1e-6, error norm over the whole world) around a
 *   right-hand side with side effects (:500-550, :437-460): trial states are written with the speed clamp / angle wrap, goal
 *   lists rotate at trial positions, the single-agent force functions of forces.py are used (with the pair force of the
 *   lower index mirrored onto the higher one under CS_ALL_PARAMS_EQUAL, forces.py:130-151).
 *   type 0..8; rows <= 64.  The robot row (CS_ROBOT_ROW) is a fixed entity for the whole call.
 *   d_memory [W][n][2]: agent.desired_force between calls (kept within one radius of the goal, forces.py:12-16); zero it
 *   when the humans are created.  d_nfev [W] or NULL: number of right-hand-side evaluations (2 + 6 per attempted step).
 *   Updates rows (x, y, yaw, Vx, Vy, BVx, BVy, Omega, goal columns) and rotates d_goals in place.
 */
int cs_update_humans_rk45(const cs_worlds* w, float dt, float* d_memory, int32_t* d_nfev, void* stream);
/*   With CS_RESPAWN the parallel-traffic respawn rule runs behind the solve, as update_humans(post_update=True) does on the agent
 *   objects (motion_model_manager.py:405-422, the non-parallel form: positions and goal lists; max over radius + safety space).
 *
 * cs_complete_rk45_simulation  replaces MotionModelManager.complete_rk45_simulation(t, dt, final_time) (motion_model_manager.py:461-498):
 *   ONE solve_ivp(f_rk45_*, (t, t + final_time), y0, method='RK45', t_eval=np.arange(t, final_time, dt)) per world -- the same
 *   right-hand side and step-size control as cs_update_humans_rk45, the solution at the n_eval = len(arange(t, final_time, dt)) times
 *   k * dt taken from scipy's RK45 dense output (rk.py RkDenseOutput: y_old + h Q [x, x^2, x^3, x^4], Q = K^T P) of the step that
 *   contains them.  d_human_states [W][n_eval][n][6] (x, y, yaw, BVx, BVy, Omega: headed types) or [..][4] (x, y, Vx, Vy): the raw
 *   solution components, as the reference returns them.  The rows of w->d_state are left at the state of t + final_time.
 *
 * cs_robot_model_rk45  replaces MotionModelManager.update_robot(t, dt) for a robot whose SFM / HSFM motion model was set with
 *   runge_kutta=True (motion_model_manager.py:631-640, f_rk45_robot_* :661-687): scipy's RK45 around compute_robot_forces (:591-613,
 *   the single-agent force functions; the humans stand still during the solve).  Arguments as cs_robot_model_step; robot_type 0..8;
 *   up to 64 humans per world.  d_nfev [W] or NULL.
 */
int cs_complete_rk45_simulation(const cs_worlds* w, float dt, float final_time, float* d_memory, float* d_human_states, int n_eval,
                                int32_t* d_nfev, void* stream);
int cs_robot_model_rk45(const cs_worlds* w, int32_t robot_type, const float* robot_params, float robot_margin, const float* d_human_margin,
                        float* d_robot_memory, float dt, int32_t* d_nfev, void* stream);

/*
 * cs_gym_bookkeeping  the host bookkeeping of SocialNavGym between two steps, for W worlds on the device (one lane per world):
 *   typed copies of cs_collision_reward's rows (d_out [W][7] -> d_reward [W], d_terminated / d_truncated [W] bytes 0 / 1,
 *   d_info [W]); the per-world step counter and global_time = clock[counter] (social_nav_gym.py:244 accumulates time_step in
 *   float32: the caller tabulates those sums in d_clock [clock_len]); with auto_reset, d_mask [W] = episode ended, the counter
 *   of such a world restarts at 0 and its seed moves on by seed_stride (the next unused seed of its arithmetic sequence) -- the inputs
 *   of the masked cs_generate_worlds that follows.
 */
int cs_gym_bookkeeping(int W, const float* d_out, int32_t* d_counter, uint32_t* d_seeds, int32_t* d_mask, float* d_global_time,
                       const float* d_clock, int clock_len, int auto_reset, float* d_reward, uint8_t* d_terminated,
                       uint8_t* d_truncated, int32_t* d_info, uint32_t seed_stride /* 0 = W */, void* stream);

/*
 * cs_gym_observe  replaces SocialNavGym.compute_humans_observable_state (social_nav_gym.py:100-105: the list of
 *   human.get_observable_state() / get_observable_state_headed(), agent.py:247-253) for W worlds: d_obs [W][n][5] rows
 *   px, py, vx, vy, radius, or [W][n][7] with theta, omega appended when theta_and_omega_visible.
 * cs_copy_worlds_masked  the second half of a masked auto-reset: the worlds with d_mask[w] != 0 are copied from `src` (a staging
 *   batch cs_generate_worlds filled, possibly on another stream beside cs_step) over those of `dst`: state rows, goal lists,
 *   robot rows, world flags.  Both descriptors have the same shape and layout.
 * With cs_collision_reward, cs_gym_bookkeeping, cs_step and cs_generate_worlds these make a whole vectorised Gym step a
 * sequence of library launches with fixed arguments: capture it once (cs_graph_begin_capture), replay it per step.
 */
int cs_gym_observe(const cs_worlds* w, int theta_and_omega_visible, float* d_obs, void* stream);
int cs_copy_worlds_masked(const cs_worlds* src, const cs_worlds* dst, const int32_t* d_mask, void* stream);
/* the same with the generator's per-world status (cs_generate_worlds d_status): a world whose generation failed (status != 0: the
 * bounded rejection sampling gave up, or the traffic is too dense) is NOT copied over the live one -- the caller sees the status. */
int cs_copy_worlds_masked_status(const cs_worlds* src, const cs_worlds* dst, const int32_t* d_mask, const int32_t* d_status, void* stream);

/*
 * cs_step_observe  cs_step followed by cs_gym_observe in ONE launch: the SFM / HSFM step kernels write the observation of the stepped
 *   humans from their registers (other crowd models and worlds beyond one block: the two launches).  d_obs as in cs_gym_observe.
 * cs_copy_worlds_masked_observe  cs_copy_worlds_masked_status that also rewrites the observation rows of the worlds it copies (d_obs
 *   [W][n][5|7], NULL: plain copy) -- with cs_step_observe a vectorised Gym step with auto-reset needs no separate observation launch.
 */
int cs_step_observe(const cs_worlds* w, float dt, int n_substeps, const float* d_action, int theta_and_omega_visible, float* d_obs,
                    void* stream);
int cs_copy_worlds_masked_observe(const cs_worlds* src, const cs_worlds* dst, const int32_t* d_mask, const int32_t* d_status,
                                  int theta_and_omega_visible, float* d_obs, void* stream);

/*
 * cs_gym_bookkeeping_next_step  cs_gym_bookkeeping for Gymnasium's NEXT_STEP autoreset mode: a world whose episode ended in the
 *   previous step (d_prev_mask[w] != 0) spends this step being reset -- its results are those of a reset step (reward 0, not
 *   terminated, not truncated, Nothing), its step counter and clock restart -- and a world that ends now is flagged in d_mask
 *   and moves to the next seed of its sequence; its replacement can then be generated BESIDE the next step (cs_generate_worlds
 *   into a staging batch on another stream) and copied in at that step's end (cs_copy_worlds_masked), off the critical path.
 */
int cs_gym_bookkeeping_next_step(int W, const float* d_out, int32_t* d_counter, uint32_t* d_seeds, int32_t* d_mask, const int32_t* d_prev_mask,
                                 float* d_global_time, const float* d_clock, int clock_len, float* d_reward, uint8_t* d_terminated,
                                 uint8_t* d_truncated, int32_t* d_info, uint32_t seed_stride /* 0 = W */, void* stream);

/*
 * cs_collision_reward_gym  cs_collision_reward and the episode bookkeeping behind it in ONE launch (the lane that writes a world's
 *   reward row also does that world's bookkeeping): what cs_collision_reward + cs_gym_bookkeeping (book->d_prev_mask == NULL) or
 *   cs_collision_reward + cs_gym_bookkeeping_next_step (d_prev_mask given) leave behind, bit for bit, with one graph node less on
 *   the critical path of a vectorised Gym step (a dependent kernel node costs ~8 us in a HIP graph replay).  d_global_time is
 *   read (time-limit test) and then advanced.  Worlds of more than 64 humans take the two launches.
 */
typedef struct cs_gym_book {
    int32_t* d_counter;          /* [W] steps of the running episode */
    uint32_t* d_seeds;           /* [W] seed of the running episode: += seed_stride when the episode ends */
    int32_t* d_mask;             /* [W] out: episode ended in this step */
    const int32_t* d_prev_mask;  /* [W] NEXT_STEP mode: worlds being reset during this step; NULL: same-step rules */
    const float* d_clock;        /* [clock_len] float32 sums of the time step */
    int32_t clock_len;
    int32_t auto_reset;          /* same-step rules: restart counter / advance seed of finished worlds */
    float* d_reward;             /* [W] typed copies of the reward row */
    uint8_t* d_terminated;
    uint8_t* d_truncated;
    int32_t* d_info;
    uint32_t seed_stride;        /* what a finished world's seed moves on by: the number of worlds of the WHOLE job, so that world w walks
                                  * s + w + k * total on any rank count (ShardedBatchedSocialNavGym); 0 = W, the single-process batch */
} cs_gym_book;
int cs_collision_reward_gym(const cs_worlds* w, const float* d_action, float T, float* d_global_time,
                            const float* reward_cfg /* host, 5 floats */, float* d_out, const cs_gym_book* book, void* stream);

/*
 * cs_gym_step  cs_collision_reward_gym followed by cs_step_observe -- the head and the body of SocialNavGym.step (social_nav_gym.py:
 *   227-250: reward / termination of the CURRENT state, then the substeps, then the observation) -- in ONE launch: the step kernel's
 *   prologue computes the swept robot-human distances of the incoming rows, writes the reward row and does the episode bookkeeping
 *   (the very code of cs_collision_reward_gym, bit for bit), then runs the fused substeps and writes the observation from its registers.
 *   One launch and one graph node less on the critical path of a vectorised Gym step.  SFM / HSFM worlds of up to 64 rows on the LDS
 *   kernel; every other world: the two launches, internally.  Arguments as the two calls'.
 */
int cs_gym_step(const cs_worlds* w, float dt, int n_substeps, const float* d_action, float T, float* d_global_time,
                const float* reward_cfg /* host, 5 floats */, float* d_out, const cs_gym_book* book, int theta_and_omega_visible,
                float* d_obs, void* stream);

/*
 * Pre-staged episodes: the reset of a finished world (SocialNavGym.reset, social_nav_gym.py:120-225: seed -> generator -> rows) taken
 * off the critical path of a vectorised Gym step.  A world's episodes are a function of their seeds, and the seeds are known ahead:
 * episode e of world w draws d_base_seed[w] + e * seed_stride.  So every world keeps its next `depth` episodes generated ahead in a
 * staging batch of depth * W worlds (slot j of world w = staging world j * W + w holds the one episode e in (epoch, epoch + depth] with
 * e = j mod depth); an episode end is then a copy, and the consumed slots are regenerated on a side stream, several steps later, without
 * anybody waiting for them -- a world that ends a few steps after its reset (a robot driven into its neighbour) finds the following
 * episodes staged as well.  No event orders the two streams: the slots carry tags.
 *   d_staged_seed[j][w]  the seed slot j of world w was generated from -- stored LAST by the generator (release, device scope)
 *   d_epoch[w]           the episode the live world w is in (0 after reset) -- stored LAST by the consumer, once every load of the slot
 *                        has returned (relaxed, device scope; the obligations of the protocol are listed in csrc/generate.hip)
 * cs_refill_staged_worlds   (side stream) regenerates every slot whose tag differs from the seed of the episode that belongs in it now
 *   (one wavefront per slot; the other blocks leave at once); d_staged_status[j][w] = cs_generate_worlds' status.  Launches of it must
 *   be ordered among themselves (one stream).  n <= 64.
 * cs_consume_staged_worlds  (the step's stream, behind the substeps) for every world with d_mask[w] != 0: episode e = epoch + 1; if the
 *   tag of slot e mod depth (a device-scope relaxed load; the slot's words are read with device-scope loads under a control dependency
 *   on it) equals d_seeds[w] -- the seed the bookkeeping just moved the world to -- the staged world is copied
 *   over the live one, observation rows included (cs_copy_worlds_masked_observe); if the refill has not got to it yet the world is
 *   generated in place from d_seeds[w] by the same generator code: the same rows either way, a function of the seed.
 *   A world that cannot be generated (status != 0) keeps its rows and gets d_failed[w] = 1 (0 after a successful reset): it starts a new
 *   episode FROM its finished rows (step counter and clock were reset by the bookkeeping) -- after a collision or a reached goal it ends
 *   again at the next step and then tries the following seed of its sequence; after a time-limit truncation it runs a further episode
 *   from where it stands, flagged in d_failed the whole time.
 */
typedef struct cs_stage_book {
    const uint32_t* d_seeds;      /* [W] cs_gym_book.d_seeds: == d_base_seed + d_epoch * seed_stride between two steps */
    const uint32_t* d_base_seed;  /* [W] seed of episode 0 */
    uint32_t* d_epoch;            /* [W] */
    uint32_t* d_staged_seed;      /* [depth][W] */
    int32_t* d_staged_status;     /* [depth][W] */
    int32_t* d_failed;            /* [W] 0 = the last take-over succeeded, 1 = the staged episode could not be generated, 2 = deferred (cs_gym_step_staged) */
    uint32_t seed_stride;         /* 0 = W */
    int32_t depth;                /* episodes staged ahead per world: a power of two */
    int32_t* d_pending;           /* [W] cs_gym_step_staged only (may be NULL otherwise): 1 = the world's episode is over and its take-over deferred */
} cs_stage_book;
/* `staging` describes depth * W worlds of the shape of the live batch (same n, G, layout, robot row) */
int cs_refill_staged_worlds(const cs_generator* gen, const cs_worlds* staging, const cs_stage_book* book, void* stream);
int cs_consume_staged_worlds(const cs_generator* gen, const cs_worlds* staging, const cs_worlds* live, const int32_t* d_mask,
                             const cs_stage_book* book, int theta_and_omega_visible, float* d_obs, void* stream);

/* cs_gym_step followed by cs_consume_staged_worlds in ONE launch (SocialNavGym.step + the reset of the worlds that ended,
 * social_nav_gym.py:227-250, :120-225): the head of the step kernel decides which worlds take over their staged episode at the end of this
 * launch -- same-step rules (book->auto_reset): the worlds that end now; NEXT_STEP rules (book->d_prev_mask): those that ended in the
 * previous step -- and the wavefront that stepped such a world copies the staged episode over it (rows, goal lists, robot row, world
 * flag, observation rows) instead of writing the stepped rows.  Same results as the two launches, bit for bit, whenever the slot is staged.
 * A slot the refill has NOT staged yet is not generated in place (cs_consume_staged_worlds does that): the take-over is DEFERRED --
 * d_pending[w] = 1, d_failed[w] = 2, the world is between two episodes (reward 0, no flags, observation of its finished rows) until a later
 * cs_gym_step_staged finds the slot.  With the refill cadence of social_gym/social_nav_gym.py (a pass whenever a world may have ended, `depth`
 * episodes ahead) that does not happen.  Only for worlds cs_gym_step runs as one launch (cs_gym_step_is_one_launch: SFM / HSFM worlds of
 * up to 64 rows on the one-wavefront builds) and without polygon walls; CS_ERR_ARG otherwise -- take the two launches there.
 * cs_gym_step_is_one_launch: 0 = cs_gym_step is two launches for these worlds, 1 = one launch, 2 = one launch and cs_gym_step_staged too. */
int cs_gym_step_is_one_launch(const cs_worlds* w);
int cs_gym_step_staged(const cs_worlds* w, float dt, int n_substeps, const float* d_action, float T, float* d_global_time, const float* reward_cfg,
                       float* d_out, const cs_gym_book* book, int theta_and_omega_visible, float* d_obs, const cs_generator* gen,
                       const cs_worlds* staging, const cs_stage_book* stage_book, void* stream);

/* layout conversion of a state array between the reference's AoS rows and SoA planes */
int cs_state_aos_to_soa(const float* d_aos, float* d_soa, int W, int rows, void* stream);
int cs_state_soa_to_aos(const float* d_soa, float* d_aos, int W, int rows, void* stream);

/* kernel launch geometry the library would use for `w` (diagnostics / DESIGN.md numbers) */
int cs_launch_geometry(const cs_worlds* w, int* grid, int* block, int* worlds_per_block);

/* Name of the kernel build the library runs for `w` (diagnostics; the parity tests assert through it that the build a
 * published number comes from is the build they compared with the oracle).  entry: 0 = cs_step, 1 = cs_update_humans_parallel
 * with d_out != d_state, 2 = cs_peek.  buf receives e.g. "k_sfm_step<SOC=0,HEADED=1,PEQ=1,MAXT=64,OCC=1,ROWS_CT=25,LEAN=1> grid=2048 block=64 wpb=2
 * lds=10128 wg=4" (lds: the dynamic LDS of a block in bytes -- what decides how many blocks a CU holds; wg: how many one-wavefront blocks go to the
 * dispatcher as one workgroup; the ORCA builds also name their arithmetic, "math=exact|fast|fma"). */
int cs_step_variant(const cs_worlds* w, int entry, char* buf, size_t buflen);

/* What CS_ORCA_MATH_DEFAULT resolves to in this process (CS_ORCA_MATH_EXACT unless CROWDSTEP_ORCA_MATH=fast|fma is set; read once). */
int cs_orca_default_math(void);

/* Diagnostic: the ORCA kernels' correctly rounded divide / square root sequences (csrc/orca.hip ieee_div, ieee_sqrt: the
 * compiler's FMA sequences without the exponent-range handling) against the compiler's operators on n_pairs random operand
 * pairs of the linear programmes' range.  h_out[0], h_out[1] = number of quotients / roots that differ in any bit (must be 0),
 * h_out[2], h_out[3] = bit patterns of the first differing operand pair.  Synchronises. */
int cs_debug_divsqrt_check(unsigned long long n_pairs, unsigned seed, unsigned long long* h_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CROWDSTEP_H */
