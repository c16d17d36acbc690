// Run-time options (options.h): a table of named integer switches, initialised once from CURLA_<NAME>, changed through
// curla_set_option().  Reads are relaxed atomics: a launch sees either the old or the new value, never a torn one, and
// nothing calls getenv() after the first use (getenv next to a setenv from another thread is a data race).
#include "options.h"

#include <atomic>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/curla_hip.h"

namespace {

struct OptDef {
  const char* name;           // curla_set_option's name; the environment variable is CURLA_ + upper case
  const char* values[6];      // value i of the option has the text values[i]; alias[i] is accepted as well
  const char* alias[6];
};

const OptDef kDefs[kOptCount] = {
    {"conv1_u8", {"auto", "hybrid", "band", "rw", "rwb", nullptr}, {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}},
    {"conv1_f32", {"rw", "band", nullptr}, {nullptr, nullptr, nullptr}},
    {"s1_fwd", {"auto", "f23", "f43", "b3", nullptr}, {nullptr, nullptr, nullptr, "bf16x3", nullptr}},
    {"bwd_split", {"auto", "0", "1", nullptr}, {nullptr, "off", "on", nullptr}},
    {"gemm_tile", {"auto", "6464", "6432", "3232", "12864", nullptr}, {nullptr, "64x64", "64x32", "32x32", "128x64", nullptr}},
    {"linear_bwd", {"pair", "split", nullptr}, {nullptr, nullptr, nullptr}},
    {"gemm_mfma", {"auto", "f32", "b3", nullptr}, {nullptr, nullptr, "bf16x3", nullptr}},
    {"s1_wgrad", {"auto", "x", "xy", nullptr}, {nullptr, "1d", "2d", nullptr}},
    {"wgrad1_u8", {"auto", "f32", "b16", nullptr}, {nullptr, nullptr, "bf16", nullptr}},
};

std::atomic<int> g_value[kOptCount];
std::once_flag g_once;

int parse(const OptDef& d, const char* text) {
  for (int i = 0; i < 6 && d.values[i]; ++i)
    if (!strcmp(text, d.values[i]) || (d.alias[i] && !strcmp(text, d.alias[i]))) return i;
  return -1;
}

void init_from_env() {
  for (int id = 0; id < kOptCount; ++id) {
    char env[64] = "CURLA_";
    size_t n = strlen(env);
    for (const char* p = kDefs[id].name; *p && n + 1 < sizeof(env); ++p) env[n++] = (char)toupper((unsigned char)*p);
    env[n] = 0;
    const char* e = getenv(env);
    const int v = e ? parse(kDefs[id], e) : 0;
    g_value[id].store(v < 0 ? 0 : v, std::memory_order_relaxed);
  }
}

int find(const char* name) {
  if (!name) return -1;
  for (int id = 0; id < kOptCount; ++id)
    if (!strcmp(name, kDefs[id].name)) return id;
  return -1;
}

}  // namespace

int curla_opt(int id) {
  std::call_once(g_once, init_from_env);
  return g_value[id].load(std::memory_order_relaxed);
}

extern "C" {

int curla_set_option(const char* name, const char* value) {
  std::call_once(g_once, init_from_env);
  const int id = find(name);
  if (id < 0 || !value) return CURLA_ERR_ARG;
  const int v = parse(kDefs[id], value);
  if (v < 0) return CURLA_ERR_ARG;
  g_value[id].store(v, std::memory_order_relaxed);
  return CURLA_OK;
}

const char* curla_get_option(const char* name) {
  std::call_once(g_once, init_from_env);
  const int id = find(name);
  return id < 0 ? nullptr : kDefs[id].values[g_value[id].load(std::memory_order_relaxed)];
}

}  // extern "C"
