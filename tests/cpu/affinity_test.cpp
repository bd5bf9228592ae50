// CPU test of csrc/twx_affinity.h: cpulist parsing, sysfs look-up against a fake tree (argv[1] = its root), binding the calling thread.
#include <assert.h>
#include <stdio.h>
#include <thread>
#include "twx_affinity.h"

int main(int argc, char** argv) {
    using namespace twx;
    std::vector<int> v;
    assert(parse_cpulist("0-3,8,10-11\n", v) && v == (std::vector<int>{0, 1, 2, 3, 8, 10, 11}));
    assert(parse_cpulist("5", v) && v == std::vector<int>{5});
    assert(parse_cpulist("", v) && v.empty());
    assert(parse_cpulist("\n", v) && v.empty());
    assert(!parse_cpulist("3-1", v) && v.empty());
    assert(!parse_cpulist("a", v) && !parse_cpulist("1,,2", v) && !parse_cpulist("1-", v) && !parse_cpulist("-1", v) && !parse_cpulist(nullptr, v));
    assert(!parse_cpulist("0-99999999999", v) && v.empty());
    if (argc > 1) {
        DeviceAffinity a = affinity_of_pci("0000:C1:00.0", argv[1]);          // HIP prints upper case on some stacks
        assert(a.numa_node == 1 && a.pci == "0000:c1:00.0" && a.cpulist == "0-1" && a.cpus == (std::vector<int>{0, 1}));
        DeviceAffinity b = affinity_of_pci("0000:05:00.0", argv[1]);          // numa_node -1 (single-node host / VM)
        assert(b.numa_node == -1 && b.cpus.size() == 1);
        DeviceAffinity c = affinity_of_pci("0000:ff:00.0", argv[1]);          // not there
        assert(c.numa_node == -1 && c.cpus.empty() && c.cpulist.empty());
        DeviceAffinity d = affinity_of_pci("0000:06:00.0", argv[1]);          // garbage in local_cpulist
        assert(d.numa_node == 0 && d.cpus.empty());
    }
    // binding: a fresh thread narrows itself to CPU 0 (always allowed here?) — only CPUs of the current mask are ever kept
    cpu_set_t cur;
    CPU_ZERO(&cur);
    assert(sched_getaffinity(0, sizeof(cur), &cur) == 0);
    int first = -1, count = 0;
    for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &cur)) { if (first < 0) first = c; ++count; }
    assert(first >= 0);
    int got = -2, after = -1;
    std::thread([&]() {
        got = pin_current_thread({first, 100000, -3});
        cpu_set_t now; CPU_ZERO(&now);
        sched_getaffinity(0, sizeof(now), &now);
        after = CPU_COUNT(&now);
    }).join();
    assert(got == 1 && after == 1);
    assert(pin_current_thread({}) == 0);
    // a list with no CPU this process may use changes nothing
    std::vector<int> none;
    for (int c = 0; c < CPU_SETSIZE; ++c) if (!CPU_ISSET(c, &cur)) { none.push_back(c); break; }
    if (!none.empty()) assert(pin_current_thread(none) == 0);
    cpu_set_t again; CPU_ZERO(&again);
    sched_getaffinity(0, sizeof(again), &again);
    assert(CPU_COUNT(&again) == count);                                          // the main thread was never narrowed
    printf("affinity ok (%d cpus)\n", count);
    return 0;
}
