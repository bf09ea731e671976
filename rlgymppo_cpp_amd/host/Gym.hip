// Gym.hip -- the standalone Gym (include/RLGymSim_CPP/Gym.h) and Arena::Step of the host facade (include/RLGymSim_CPP/RocketSim/Arena.h)
// over a one-env device batch.  What it restates: SIM/Gym.cpp:40-102 (constructor, Reset, Step) with the arena work done by the step
// kernel behind include/rlgpu.h and the match's plugins called on the host in the reference's order.  Off the hot path: one env, a state
// upload and two downloads per step.
#include "host_util.h"

namespace {
template <class T>
T* dev_alloc(size_t n) { T* p = nullptr; HOST_HIP(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T))); return p; }

// the arena's one-env batch; recreated when a Gym asks for another tick skip than the arena was created with
struct ArenaDevice {
    rlgpu_env* env = nullptr; RlgpuGymConfig cfg{}; int teamSize = 0;
    ~ArenaDevice() { if (env) rlgpu_env_destroy(env); }
};
void EnvCheck(rlgpu_env* env, int rc, const char* what) { if (rc != RLGPU_OK) RG_ERR_CLOSE("rlgpu_env_" << what << " failed (" << rc << "): " << rlgpu_env_last_error(env)); }

ArenaDevice* EnsureDevice(RocketSim::Arena* arena, RlgpuGymConfig cfg) {
    void*& slot = arena->_device;
    ArenaDevice* d = static_cast<ArenaDevice*>(slot);
    const int cars = (int)arena->_cars.size();
    if (cars == 1 && !arena->_IsOneTeam()) arena->_SetOneTeam();   // a lone blue car: an arena without opponents
    cfg.one_team = arena->_IsOneTeam() ? 1 : 0;
    const int teamSize = cfg.one_team ? cars : cars / 2;
    if (teamSize < 1 || teamSize > 3 || (!cfg.one_team && cars % 2)) RG_ERR_CLOSE("Arena: the device stepper runs 1 to 3 cars per team, with or without opponents (got " << cars << " cars)");
    if (d && (std::memcmp(&d->cfg, &cfg, sizeof(cfg)) != 0 || d->teamSize != teamSize)) { delete d; d = nullptr; }
    if (!d) {
        d = new ArenaDevice(); d->cfg = cfg; d->teamSize = teamSize;
        int rc = rlgpu_env_create(&d->env, 0, 1, d->teamSize, &cfg);
        if (rc != RLGPU_OK) { delete d; RG_ERR_CLOSE("rlgpu_env_create failed (" << rc << ")"); }
        RLGSC::LoadArenaMesh(d->env, true);
        EnvCheck(d->env, rlgpu_env_enable_snapshots(d->env, 1), "enable_snapshots");
    }
    slot = d;
    return d;
}
// neutral gym settings: everything a Match decides is decided on the host
RlgpuGymConfig NeutralConfig(int tickSkip) {
    RlgpuGymConfig cfg; rlgpu_default_gym_config(&cfg);
    cfg.tick_skip = tickSkip; cfg.n_terms = 0; cfg.n_conds = 0; cfg.zero_sum = 0; cfg.setter_kind = RLGPU_SS_KICKOFF;
    for (int i = 0; i < RLGPU_NUM_EVENT_VALS; i++) cfg.event_weights[i] = 0.f;
    return cfg;
}
}  // namespace

namespace RocketSim {
void Arena::Step(int ticksToSimulate) {
    if (ticksToSimulate <= 0) return;
    ArenaDevice* d = _device ? static_cast<ArenaDevice*>(_device) : EnsureDevice(this, NeutralConfig(8));
    _SyncToState();
    EnvCheck(d->env, rlgpu_env_upload_states(d->env, &_state, nullptr, 1), "upload_states");
    if (!_goalScoreCallback.func && !_carBumpCallback.func) {
        EnvCheck(d->env, rlgpu_env_physics_ticks(d->env, ticksToSimulate), "physics_ticks");
        EnvCheck(d->env, rlgpu_env_download_states(d->env, &_state, nullptr, 1), "download_states");
        _SyncFromState();
        return;
    }
    // With a callback set the arena is read back after every tick and the events of that tick are raised from what the kernel left in
    // the state: a bump (Arena.cpp:335-413) is the one thing that raises a car's carContact cooldown or changes whom it is against, and
    // it was a demo when the victim went from alive to demoed in that tick; the goal callback fires on every tick that ends with the
    // ball behind a goal line (Arena.cpp:804-808).  The cars' order is the reference's (callbacks of one tick arrive bumper by bumper).
    for (int t = 0; t < ticksToSimulate; t++) {
        const RlgpuArenaState before = _state;
        EnvCheck(d->env, rlgpu_env_physics_ticks(d->env, 1), "physics_ticks");
        EnvCheck(d->env, rlgpu_env_download_states(d->env, &_state, nullptr, 1), "download_states");
        _SyncFromState();
        if (_carBumpCallback.func)
            for (int k = 0; k < _state.num_cars; k++) {
                const RlgpuCarState &was = before.cars[k], &now = _state.cars[k];
                if ((now.flags & RLGPU_CF_ABSENT) || now.car_contact_other_id <= 0 || !(now.car_contact_cooldown > 0)) continue;
                const bool fresh = now.car_contact_cooldown > was.car_contact_cooldown || (now.car_contact_other_id != was.car_contact_other_id && now.car_contact_cooldown >= was.car_contact_cooldown);
                if (!fresh) continue;
                const int v = now.car_contact_other_id - 1;
                Car *bumper = _CarOfSlot(k), *victim = _CarOfSlot(v);
                if (!bumper || !victim) continue;
                const bool isDemo = (_state.cars[v].flags & RLGPU_CF_IS_DEMOED) && !(before.cars[v].flags & RLGPU_CF_IS_DEMOED);
                _carBumpCallback.func(this, bumper, victim, isDemo, _carBumpCallback.userInfo);
            }
        if (_goalScoreCallback.func && IsBallScored())
            _goalScoreCallback.func(this, -_state.ball.pos[1] < 0 ? Team::BLUE : Team::ORANGE, _goalScoreCallback.userInfo);   // RS_TEAM_FROM_Y(-y), Car.h:131
    }
}
void Arena::ReleaseDevice() { delete static_cast<ArenaDevice*>(_device); _device = nullptr; }
}  // namespace RocketSim

namespace RLGSC {

Arena* MakeScratchArena(int teamSize, bool spawnOpponents) {
    Arena* a = Arena::Create(GameMode::SOCCAR);
    for (int i = 0; i < teamSize; i++) { a->AddCar(Team::BLUE); if (spawnOpponents) a->AddCar(Team::ORANGE); }
    if (!spawnOpponents) a->_SetOneTeam();
    return a;
}

struct Gym::Device {
    float *controls = nullptr, *obs = nullptr, *rew = nullptr; int32_t* done = nullptr;
    ~Device() { for (void* p : {(void*)controls, (void*)obs, (void*)rew, (void*)done}) if (p) (void)hipFree(p); }
};

Gym::Gym(Match* match, int tickSkip, CarConfig carConfig, GameMode gameMode, MutatorConfig mutatorConfig)
    : match(match), tickSkip(tickSkip), actionDelay(tickSkip - 1) {
    arena = Arena::Create(gameMode);
    arena->SetMutatorConfig(mutatorConfig);
    for (int i = 0; i < match->teamSize; i++) {
        carIds.push_back(arena->AddCar(Team::BLUE, carConfig)->id);
        if (match->spawnOpponents) carIds.push_back(arena->AddCar(Team::ORANGE, carConfig)->id);
    }
    if (!match->spawnOpponents) arena->_SetOneTeam();
}
Gym::~Gym() { delete dev; delete arena; }

// Gym.cpp:58-66
FList2 Gym::Reset() {
    const GameState setterState = match->ResetState(arena);
    // the device does the episode bookkeeping on the new state: counters, score line, event tracker (Gym.cpp:62-63)
    ArenaDevice* d = EnsureDevice(arena, NeutralConfig(tickSkip));
    arena->_SyncToState();
    const int32_t env0 = 0;
    EnvCheck(d->env, rlgpu_env_upload_states(d->env, &arena->_state, nullptr, 1), "upload_states");
    EnvCheck(d->env, rlgpu_env_reset_envs(d->env, &env0, 1, 0, nullptr), "reset_envs");
    EnvCheck(d->env, rlgpu_env_download_states(d->env, &arena->_state, nullptr, 1), "download_states");
    GameState resetState(arena);
    // the episode's first GameState is the one the state setter returned: built BEFORE Match::ResetState reset the pads (Match.cpp:55-69)
    std::copy(std::begin(setterState.boostPads), std::end(setterState.boostPads), std::begin(resetState.boostPads));
    std::copy(std::begin(setterState.boostPadsInv), std::end(setterState.boostPadsInv), std::begin(resetState.boostPadsInv));
    match->EpisodeReset(resetState);
    prevState = resetState;
    eventTracker.ResetPersistentInfo();
    return match->BuildObservations(resetState);
}

// Gym.cpp:68-102
Gym::StepResult Gym::Step(const ActionParser::Input& actionsData) {
    ActionSet actions = match->ParseActions(actionsData, prevState);
    match->prevActions = actions;
    const int P = match->playerAmount;
    if ((int)actions.size() != P) RG_ERR_CLOSE("Gym::Step(): " << actions.size() << " actions for " << P << " players");
    ArenaDevice* d = EnsureDevice(arena, NeutralConfig(tickSkip));
    if (!dev) {
        dev = new Device();
        dev->controls = dev_alloc<float>((size_t)P * 8); dev->obs = dev_alloc<float>((size_t)P * rlgpu_env_obs_size(d->env));
        dev->rew = dev_alloc<float>(P); dev->done = dev_alloc<int32_t>(P);
    }
    std::vector<float> rows((size_t)P * 8);
    for (int i = 0; i < P; i++) {
        for (int j = 0; j < 8; j++) rows[(size_t)i * 8 + j] = actions[i][j];
        CarControls& c = arena->_cars[i]->controls;   // Action -> CarControls (Action.h:36-46)
        c.throttle = actions[i].throttle; c.steer = actions[i].steer; c.pitch = actions[i].pitch; c.yaw = actions[i].yaw; c.roll = actions[i].roll;
        c.jump = actions[i].jump == 1.f; c.boost = actions[i].boost == 1.f; c.handbrake = actions[i].handbrake == 1.f;
    }
    // the arena may have been edited through the facade since the last step: it goes up whole
    arena->_SyncToState();
    EnvCheck(d->env, rlgpu_env_upload_states(d->env, &arena->_state, nullptr, 1), "upload_states");
    HOST_HIP(hipMemcpy(dev->controls, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    // arena->Step(tickSkip - actionDelay), eventTracker.Update, state.UpdateFromArena, arena->Step(actionDelay): one launch
    EnvCheck(d->env, rlgpu_env_step_controls(d->env, dev->controls, dev->obs, dev->rew, dev->done), "step_controls");
    RlgpuArenaState snap;
    EnvCheck(d->env, rlgpu_env_download_snapshots(d->env, &snap, 0, 1), "download_snapshots");
    EnvCheck(d->env, rlgpu_env_download_states(d->env, &arena->_state, nullptr, 1), "download_states");
    arena->_SyncFromState();
    GameState state(snap, (int)((uint64_t)snap.tick_count - prevState.lastTickCount));
    totalTicks += tickSkip; totalSteps++;

    StepResult result;
    result.obs = match->BuildObservations(state);
    result.done = match->IsDone(state);
    result.reward = match->GetRewards(state, result.done);
    prevState = state;
    result.state = state;
    return result;
}

}  // namespace RLGSC
