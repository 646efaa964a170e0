// qg_device.h -- constant tables and launch parameters shared by the host side
// of the C ABI and the gfx950 kernels (single precision, kernarg-resident).
#pragma once
#include <stdint.h>

#define QGK_NLINK 12       // fema / shin / foot x 4 legs
#define QGK_CP_FRAME 12    // contact sample points on the FRAME
#define QGK_CP_LINK 8      // ... on every leg link
#define QGK_WAVE 64

// One leg link and the hinge that drives it (quadruped.xml:71-141, joint defaults :9,24-37,
// servo defaults :10-37).  All hinge axes are the link's local z (quadruped.xml:9), which the
// host side verifies before building this table.
struct KLink {
    float pos[3];       // link frame origin in the parent frame
    float Q[9];         // link frame orientation in the parent frame, row-major rotation matrix
    float mass;
    float ipos[3];      // centre of mass, link frame
    float inertia[6];   // xx yy zz xy xz yz about the COM, link axes
    float cp[QGK_CP_LINK][3];
    // hinge
    float ref, lo, hi, damping, armature;
    // position servo
    float kp, kv, gear, ctrl_lo, ctrl_hi, force_lo, force_hi, act_decay;  // act_decay = 1 - exp(-h/timeconst)
};

struct KModel {
    float h;            // timestep
    float g[3];         // gravity, world
    // FRAME rigid inertia about its own origin, in its own axes
    float m0, h0[3], I0[6];
    float free_damping, free_armature;
    float cp0[QGK_CP_FRAME][3];
    float contact_k, contact_c, contact_margin, contact_mu, contact_inv_ramp;
    float limit_k, limit_b, limit_inv_ramp;
    float qpos0[19];
    KLink link[QGK_NLINK];
};

struct KTask {
    int32_t frame_skip;
    int32_t limit_substeps;   // substep count at which data.time >= max_time (f64 accumulation), or INT32_MAX
    int32_t use_fall;
    float fall_height;
    int32_t use_flip;         // body z axis of the (lagged) sensor pack below the horizon terminates
    float w_forward, w_ctrl, alive_bonus;
    int32_t obs_mode;         // 0: 33 sensors, 1: 21-value IMU pack
    int32_t sensor_lag;
    int32_t auto_reset;
    uint32_t reset_flags;
    float default_ctrl[12];
    float reset_joint_jitter;
};

// Struct-of-arrays state in HBM: field-major, env-minor, so that lane i of a wave
// touches address base + i*4 for every field (one 256-byte segment per wave and field).
struct KState {
    float *qpos;      // [19][n]
    float *qvel;      // [18][n]
    float *act;       // [12][n]
    float *ctrl;      // [12][n]  last env-clipped action (data.ctrl); written only when track_ctrl
    int32_t *nstep;   // [n]
    int32_t *episode; // [n]   resets this env has gone through: the counter of its reset random stream (graph-replay safe)
};

struct KStepArgs {
    KState st;
    int32_t n;
    int32_t track_ctrl;
    const float *actions;     // [n][12]
    float *obs;               // [n][obs_dim]   (separate outputs) or NULL
    float *reward;            // [n]
    uint8_t *done;            // [n]
    float *comps;             // [n][3] or NULL
    float *packed;            // [n][obs_dim+2] or NULL
    uint64_t seed;            // reset stream
    uint64_t env_index_base;
};
