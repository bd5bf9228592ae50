// twx_affinity.h — which host CPUs sit next to a GPU, and binding the calling thread to them.  Plain C++ (no HIP in this file: the
// caller passes the device's PCI bus id, hipDeviceGetPCIBusId), so the CPU test suite runs it against a fake sysfs tree
// (tests/cpu/affinity_test.cpp).
//
// The reference's concurrent correlations are worker threads inside one program (processing/CPP/main.cpp:180-187,488-497) and jobs
// side by side (acquisition/goprocess.sh:9-11); on an 8-GPU node every such thread / rank feeds ONE device over PCIe from pinned host
// memory, so it belongs on the NUMA node that device hangs off: /sys/bus/pci/devices/<bus id>/numa_node and …/local_cpulist.
#pragma once
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

namespace twx {

struct DeviceAffinity {
    int numa_node = -1;              // -1: the platform does not say (single-node hosts, VMs)
    std::string pci, cpulist;        // "0000:05:00.0", "0-31,128-159"
    std::vector<int> cpus;           // cpulist expanded
};

// "0-3,8,10-11" -> {0,1,2,3,8,10,11}; false on anything else (the list is then left empty)
inline bool parse_cpulist(const char* s, std::vector<int>& out) {
    out.clear();
    if (!s) return false;
    const char* p = s;
    while (*p == ' ' || *p == '\t') ++p;
    if (!*p || *p == '\n') return true;                      // an empty list is a valid answer (no local CPUs)
    for (;;) {
        char* e = nullptr;
        const long a = strtol(p, &e, 10);
        if (e == p || a < 0 || a > 1 << 20) { out.clear(); return false; }
        long b = a;
        p = e;
        if (*p == '-') {
            ++p;
            b = strtol(p, &e, 10);
            if (e == p || b < a || b > 1 << 20) { out.clear(); return false; }
            p = e;
        }
        for (long c = a; c <= b; ++c) out.push_back((int)c);
        if (*p == ',') { ++p; continue; }
        while (*p == ' ' || *p == '\t' || *p == '\n') ++p;
        if (*p) { out.clear(); return false; }
        return true;
    }
}

inline bool read_small_file(const std::string& path, std::string& out) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[4096];
    const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
    fclose(f);
    buf[n] = 0;
    out = buf;
    while (!out.empty() && (out.back() == '\n' || out.back() == ' ')) out.pop_back();
    return true;
}

// sysfs answers for one PCI function; bus ids are lower-cased (HIP prints "0000:C1:00.0" on some stacks, sysfs names are lower case)
inline DeviceAffinity affinity_of_pci(const std::string& busid, const char* sysroot = "/sys") {
    DeviceAffinity a;
    a.pci = busid;
    std::transform(a.pci.begin(), a.pci.end(), a.pci.begin(), [](unsigned char c) { return (char)tolower(c); });
    const std::string base = std::string(sysroot) + "/bus/pci/devices/" + a.pci + "/";
    std::string t;
    if (read_small_file(base + "numa_node", t)) {
        char* e = nullptr;
        const long v = strtol(t.c_str(), &e, 10);
        if (e != t.c_str()) a.numa_node = (int)v;
    }
    if (read_small_file(base + "local_cpulist", t) && parse_cpulist(t.c_str(), a.cpus)) a.cpulist = t;
    return a;
}

// Binds the CALLING thread to `cpus` ∩ (the CPUs it may run on now: cgroup / taskset limits are kept).  Returns the number of
// CPUs in the new mask; 0 = nothing done (empty list or empty intersection), -1 = the system call failed.
inline int pin_current_thread(const std::vector<int>& cpus) {
    if (cpus.empty()) return 0;
    cpu_set_t cur, want;
    CPU_ZERO(&cur); CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(cur), &cur) != 0) return -1;
    int n = 0;
    for (int c : cpus) if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &cur)) { CPU_SET(c, &want); ++n; }
    if (n == 0) return 0;
    return sched_setaffinity(0, sizeof(want), &want) == 0 ? n : -1;
}

}  // namespace twx
