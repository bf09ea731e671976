// RenderSender (PUB/Util/RenderSender.{h,cpp} + python_scripts/render_receiver.py): the reference serialises the rendered game's
// state to JSON, hands it to an embedded Python module, and that module forwards a RocketSimVis datagram over UDP
// (127.0.0.1:9273).  Here the datagram is built and sent natively, with the schema the receiver produces:
//   {"gamemode", "ball_phys": {pos, vel, ang_vel}, "cars": [{car_id, team_num, phys{pos,forward,right,up,vel,ang_vel},
//    boost_pickups, is_demoed, on_ground, ball_touched, has_flip, boost_amount}], "boost_pad_states": [34 x bool]}
#pragma once
#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>
#include <RLGymSim_CPP/Utils/Gamestates/GameState.h>
#include "../Framework.h"
namespace RLGPC {
struct RenderSender {
    int sock = -1;
    sockaddr_in addr{};
    uint64_t sent = 0;

    RenderSender(const char* ip = "127.0.0.1", int port = 9273) {
        RG_LOG("Initializing RenderSender...");
        sock = socket(AF_INET, SOCK_DGRAM, 0);
        if (sock < 0) RG_ERR_CLOSE("RenderSender: Failed to create the UDP socket");
        addr.sin_family = AF_INET; addr.sin_port = htons((uint16_t)port);
        if (inet_pton(AF_INET, ip, &addr.sin_addr) != 1) RG_ERR_CLOSE("RenderSender: bad address " << ip);
        RG_LOG(" > RenderSender initalized (RocketSimVis datagrams to " << ip << ":" << port << ").");
    }
    RenderSender(const RenderSender&) = delete;
    RenderSender& operator=(const RenderSender&) = delete;
    ~RenderSender() { if (sock >= 0) close(sock); }

    static void VecJson(std::ostream& o, const char* key, const RLGSC::Vec& v) { o << "\"" << key << "\": [" << v.x << ", " << v.y << ", " << v.z << "]"; }

    // the datagram for one state (RenderSender.cpp:26-92 piped through render_receiver.py:20-33)
    static std::string ToJSON(const RLGSC::GameState& state, const RLGSC::ActionSet& actions) {
        (void)actions;   // the receiver drops the actions before forwarding (render_receiver.py:20-33)
        std::ostringstream o;
        o << std::setprecision(9) << "{\"gamemode\": \"soccar\", \"ball_phys\": {";
        VecJson(o, "pos", state.ball.pos); o << ", "; VecJson(o, "vel", state.ball.vel); o << ", "; VecJson(o, "ang_vel", state.ball.angVel);
        o << "}, \"cars\": [";
        for (size_t i = 0; i < state.players.size(); i++) {
            const RLGSC::PlayerData& p = state.players[i];
            o << (i ? ", " : "") << "{\"car_id\": " << p.carId << ", \"team_num\": " << (int)p.team << ", \"phys\": {";
            VecJson(o, "pos", p.phys.pos); o << ", "; VecJson(o, "forward", p.phys.rotMat.forward); o << ", "; VecJson(o, "right", p.phys.rotMat.right); o << ", ";
            VecJson(o, "up", p.phys.rotMat.up); o << ", "; VecJson(o, "vel", p.phys.vel); o << ", "; VecJson(o, "ang_vel", p.phys.angVel);
            o << "}, \"boost_pickups\": " << p.boostPickups << ", \"is_demoed\": " << (p.carState.isDemoed ? "true" : "false") << ", \"on_ground\": "
              << (p.carState.isOnGround ? "true" : "false") << ", \"ball_touched\": " << (p.ballTouchedStep ? "true" : "false") << ", \"has_flip\": "
              << (p.hasFlip ? "true" : "false") << ", \"boost_amount\": " << p.boostFraction << "}";
        }
        o << "], \"boost_pad_states\": [";
        for (int i = 0; i < RLGSC::CommonValues::BOOST_LOCATIONS_AMOUNT; i++) o << (i ? ", " : "") << (state.boostPads[i] ? "true" : "false");
        o << "]}";
        return o.str();
    }

    void Send(const RLGSC::GameState& state, const RLGSC::ActionSet& actions) {
        const std::string j = ToJSON(state, actions);
        // like the Python receiver's sendto: fire and forget (no viewer listening is not an error)
        (void)sendto(sock, j.data(), j.size(), 0, (const sockaddr*)&addr, sizeof(addr));
        sent++;
    }
};
}
