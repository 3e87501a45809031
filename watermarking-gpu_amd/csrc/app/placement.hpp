// placement.hpp -- where a device's host thread (and, by first touch, its pinned staging buffers) should live: the CPUs of the
// NUMA node the GPU hangs off.  The C++ twin of watermarking-gpu_amd/placement.py for wm_stream's per-device worker threads.
// The buffer being placed is the reference's one host frame buffer (main.cpp:273-275), here a pinned ring per device that
// every frame crosses twice; at 8 devices x 45 GB/s each way a ring on the wrong socket is paid for on every frame.
//   /sys/bus/pci/devices/<domain:bus:dev.fn>/numa_node      -1: the platform does not say (no pinning)
//   /sys/bus/pci/devices/<domain:bus:dev.fn>/local_cpulist  "0-63,128-191"
// The PCI address is what hipDeviceGetPCIBusId reports, so no guess about device order is involved.
#pragma once
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

namespace wmplace {

inline std::vector<int> parse_cpulist(const std::string& text)
{
    std::vector<int> out;
    size_t p = 0;
    while (p < text.size()) {
        size_t q = text.find(',', p);
        if (q == std::string::npos) q = text.size();
        std::string part = text.substr(p, q - p);
        part.erase(std::remove_if(part.begin(), part.end(), [](unsigned char c) { return std::isspace(c); }), part.end());
        if (!part.empty()) {
            const size_t d = part.find('-');
            const int a = std::atoi(part.substr(0, d).c_str());
            const int b = d == std::string::npos ? a : std::atoi(part.substr(d + 1).c_str());
            for (int c = a; c <= b; ++c) out.push_back(c);
        }
        p = q + 1;
    }
    return out;
}

inline std::string format_cpulist(const std::vector<int>& cpus_in)
{
    std::vector<int> cs = cpus_in;
    std::sort(cs.begin(), cs.end());
    std::string out;
    for (size_t i = 0; i < cs.size();) {
        size_t j = i;
        while (j + 1 < cs.size() && cs[j + 1] == cs[j] + 1) ++j;
        if (!out.empty()) out += ",";
        out += i == j ? std::to_string(cs[i]) : std::to_string(cs[i]) + "-" + std::to_string(cs[j]);
        i = j + 1;
    }
    return out;
}

struct Plan {
    bool valid = false;     // sysfs names a NUMA node and at least one of its CPUs is ours
    int numa_node = -1;
    std::vector<int> cpus;  // local CPUs of the device, intersected with the CPUs this process may use
};

// `allowed`: the CPUs the process may use (empty = no restriction known)
inline Plan plan_for_pci(const std::string& sysfs_root, std::string pci, const std::vector<int>& allowed)
{
    Plan p;
    std::transform(pci.begin(), pci.end(), pci.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    const std::string dir = sysfs_root + "/bus/pci/devices/" + pci;
    std::ifstream fn(dir + "/numa_node"), fc(dir + "/local_cpulist");
    if (!fn.is_open() || !fc.is_open()) return p;
    fn >> p.numa_node;
    std::string list;
    std::getline(fc, list);
    if (!fn || p.numa_node < 0) return p;
    for (int c : parse_cpulist(list))
        if (allowed.empty() || std::find(allowed.begin(), allowed.end(), c) != allowed.end()) p.cpus.push_back(c);
    p.valid = !p.cpus.empty();
    return p;
}

inline std::vector<int> allowed_cpus()
{
    std::vector<int> out;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) != 0) return out;
    for (int c = 0; c < CPU_SETSIZE; ++c)
        if (CPU_ISSET(c, &set)) out.push_back(c);
    return out;
}

// pins the CALLING thread; threads it creates afterwards inherit the mask
inline bool apply_to_this_thread(const Plan& p)
{
    if (!p.valid) return false;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : p.cpus)
        if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
    return pthread_setaffinity_np(pthread_self(), sizeof set, &set) == 0;
}

}  // namespace wmplace
