// MetricSender (PUB/Util/MetricSender.{h,cpp}): the reference hands every iteration's Report to an embedded Python module that logs
// it to wandb (python_scripts/metric_receiver.py).  Here the sender is native: one JSON object per Send() appended to
// <RLGPU_METRICS_DIR or "metrics">/<project>/<run id>.jsonl, first line = the run description; tools/metric_receiver.py tails such
// a file into wandb with the receiver's own calls (init(project, group, name, id, resume) / log).  Same fields and run-id
// continuation (RUNNING_STATS.json "run_id", Learner.cpp:204-205,238-239) as the reference.
#pragma once
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include "Report.h"
namespace RLGPC {
struct MetricSender {
    std::string curRunID;
    std::string projectName, groupName, runName;
    std::filesystem::path filePath;

    static std::string JsonString(const std::string& s) {
        std::string o = "\"";
        for (char c : s) {
            if (c == '"' || c == '\\') { o += '\\'; o += c; }
            else if ((unsigned char)c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
            else o += c;
        }
        return o + "\"";
    }

    MetricSender(std::string projectName_ = {}, std::string groupName_ = {}, std::string runName_ = {}, std::string runID = {})
        : projectName(projectName_), groupName(groupName_), runName(runName_) {
        RG_LOG("Initializing MetricSender...");
        if (runID.empty()) {   // wandb-style 8 character id
            uint64_t x = (uint64_t)std::chrono::high_resolution_clock::now().time_since_epoch().count() * 6364136223846793005ull + 1442695040888963407ull;
            for (int i = 0; i < 8; i++) { x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ull; curRunID += "0123456789abcdefghijklmnopqrstuvwxyz"[(x >> 33) % 36]; }
        } else curRunID = runID;
        const char* dir = std::getenv("RLGPU_METRICS_DIR");
        std::filesystem::path folder = std::filesystem::path(dir && *dir ? dir : "metrics") / (projectName.empty() ? "default" : projectName);
        std::error_code ec;
        std::filesystem::create_directories(folder, ec);
        filePath = folder / (curRunID + ".jsonl");
        const bool fresh = !std::filesystem::exists(filePath);
        std::ofstream f(filePath, std::ios::app);
        if (!f.good()) RG_ERR_CLOSE("MetricSender: Failed to initialize, can't open " << filePath.string());
        if (fresh)
            f << "{\"_run\": {\"project\": " << JsonString(projectName) << ", \"group\": " << JsonString(groupName) << ", \"name\": " << JsonString(runName)
              << ", \"id\": " << JsonString(curRunID) << "}}\n";
        RG_LOG(" > " << (runID.empty() ? "Starting" : "Continuing") << " run with ID : \"" << curRunID << "\"...");
        RG_LOG(" > MetricSender initalized, writing " << filePath.string());
    }
    MetricSender(const MetricSender&) = delete;
    MetricSender& operator=(const MetricSender&) = delete;

    void Send(const Report& report) {
        std::ofstream f(filePath, std::ios::app);
        if (!f.good()) RG_ERR_CLOSE("MetricSender: Failed to add metrics, can't open " << filePath.string());
        f << std::setprecision(17) << "{";
        bool first = true;
        for (auto& pair : report.data) {
            f << (first ? "" : ", ") << JsonString(pair.first) << ": ";
            if (std::isfinite(pair.second)) f << pair.second; else f << "null";
            first = false;
        }
        f << "}\n";
    }
};
}
