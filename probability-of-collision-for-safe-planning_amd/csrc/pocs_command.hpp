// pocs_command.hpp -- the text-command grammar of the module, host only (no HIP type, no device call).
//
// Stands in for the token handling of the reference's fifteen command handlers
// (mcsimplugin/mcsimplugin.cpp:47-232): each reads doubles / ints from the command's input stream with
// operator>> and checks nothing -- `setAlphas` copies EVERY remaining token into a 1 x 4 matrix
// (:176-184 + MCSimulator.h:143,226-228: a fifth token writes out of bounds), `setLandmarks` / `setTrajectory`
// / `setOdometry` loop to counts set by EARLIER commands (:148-166, :83-113) and read garbage when those were
// never sent, eight handlers fall off the end of a bool function (:83-172).  Here a line is split, its tokens
// are parsed and COUNTED against what the command takes (for the three count-dependent commands: against the
// counts the earlier commands fixed, handed in as `Shape`), and the result is a value: command id, numbers,
// integer / seed argument, or an error code of include/pocs.h with its text.  pocs_send_command
// (pocs_host.hip) dispatches on it; nothing here touches a context, so the grammar is compiled and fuzzed on
// the CPU under AddressSanitizer / UBSan (tests/command_fuzz.cpp, tests/test_sanitizers.py).
#pragma once
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "../../include/pocs.h"

namespace pocs_cmd {

enum Id {
  kMyCommand, kArmaCommand, kHelp, kSetAlphas, kSetQ, kSetNumLandmarks, kSetLandmarks, kSetNumParticles,
  kSetInitialCovariance, kSetPathLength, kSetTrajectory, kSetOdometry, kRunSimulation, kSetNumGaussians,
  kRunGMMEstimation, kSetNumGMMSamples, kSetSeed, kSetFootprint, kAddObstacle, kClearObstacles, kSetBatch,
  kSetRunAhead, kUnknown
};

// What earlier commands have fixed: the token counts of setLandmarks / setTrajectory / setOdometry depend on it
// (-1 = not sent yet -> POCS_E_ORDER, where the reference would loop over an uninitialised count).
struct Shape {
  int num_landmarks;
  int path_length;
};

struct Parsed {
  Id id = kUnknown;
  int err = POCS_OK;              // POCS_OK or a POCS_E_* code
  std::string name, msg;          // the command's name as sent; the error text
  std::vector<double> v;          // numeric tokens (setters)
  long long n = 0;                // the integer argument of the one-integer commands
  unsigned long long seed = 0;    // setSeed
};

inline bool is_space(char ch) { return ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r'; }

// whitespace-separated doubles, all of them or none: "1 2 x" is malformed, "" is zero numbers
inline bool split_numbers(const char* s, std::vector<double>* out) {
  out->clear();
  while (*s) {
    while (is_space(*s)) ++s;
    if (!*s) break;
    char* end = nullptr;
    const double v = strtod(s, &end);
    if (end == s) return false;
    if (*end && !is_space(*end)) return false;
    out->push_back(v);
    s = end;
  }
  return true;
}

inline bool is_integer(double v) { return v == floor(v) && fabs(v) < 9.0e15; }

inline Id lookup(const std::string& name) {
  static const struct { const char* name; Id id; } kTable[] = {
      {"MyCommand", kMyCommand}, {"ArmaCommand", kArmaCommand}, {"help", kHelp}, {"setAlphas", kSetAlphas}, {"setQ", kSetQ},
      {"setNumLandmarks", kSetNumLandmarks}, {"setLandmarks", kSetLandmarks}, {"setNumParticles", kSetNumParticles},
      {"setInitialCovariance", kSetInitialCovariance}, {"setPathLength", kSetPathLength}, {"setTrajectory", kSetTrajectory},
      {"setOdometry", kSetOdometry}, {"runSimulation", kRunSimulation}, {"setNumGaussians", kSetNumGaussians},
      {"runGMMEstimation", kRunGMMEstimation}, {"setNumGMMSamples", kSetNumGMMSamples}, {"setSeed", kSetSeed},
      {"setFootprint", kSetFootprint}, {"addObstacle", kAddObstacle}, {"clearObstacles", kClearObstacles},
      {"setBatch", kSetBatch}, {"setRunAhead", kSetRunAhead}};
  for (const auto& e : kTable) if (name == e.name) return e.id;
  return kUnknown;
}

inline Parsed& fail(Parsed& p, int code, const char* fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  p.err = code;
  p.msg = buf;
  return p;
}

// "<name> <tokens...>" -> Parsed.  `line` must be NUL-terminated; nothing else is assumed about it.
inline Parsed parse(const char* line, const Shape& shape) {
  Parsed p;
  if (!line) return fail(p, POCS_E_ARG, "null command line");
  while (*line == ' ' || *line == '\t') ++line;
  const char* sp = line;
  while (*sp && !is_space(*sp)) ++sp;
  p.name.assign(line, sp);
  const char* rest = sp;
  p.id = lookup(p.name);
  const char* nm = p.name.c_str();
  auto numbers = [&](size_t want) -> bool {
    if (!split_numbers(rest, &p.v)) { fail(p, POCS_E_ARG, "%s: malformed number", nm); return false; }
    if (p.v.size() != want) { fail(p, POCS_E_ARG, "%s: expected %zu values, got %zu", nm, want, p.v.size()); return false; }
    return true;
  };
  auto one_int = [&]() -> bool {
    if (!numbers(1)) return false;
    if (!is_integer(p.v[0])) { fail(p, POCS_E_ARG, "%s: integer expected", nm); return false; }
    p.n = (long long)p.v[0];
    return true;
  };
  switch (p.id) {
    case kMyCommand: case kArmaCommand: case kHelp: case kClearObstacles: case kRunSimulation: case kRunGMMEstimation:
      break;                                                      // take no tokens; whatever follows is ignored, as by the reference
    case kSetAlphas:                                              // mcsimplugin.cpp:174-187: every remaining token; more than four overflowed there
      if (!split_numbers(rest, &p.v)) return fail(p, POCS_E_ARG, "setAlphas: malformed number");
      if (p.v.empty() || p.v.size() > 4) return fail(p, POCS_E_ARG, "setAlphas takes 1..4 values (got %zu)", p.v.size());
      break;
    case kSetQ: numbers(1); break;
    case kSetNumLandmarks: case kSetNumParticles: case kSetPathLength: case kSetNumGaussians: case kSetNumGMMSamples:
    case kSetBatch: case kSetRunAhead:
      one_int();
      break;
    case kSetLandmarks:
      if (shape.num_landmarks < 0) return fail(p, POCS_E_ORDER, "setLandmarks before setNumLandmarks");
      numbers((size_t)2 * (size_t)shape.num_landmarks);
      break;
    case kSetInitialCovariance: numbers(9); break;
    case kSetTrajectory:
      if (shape.path_length < 1) return fail(p, POCS_E_ORDER, "setTrajectory before setPathLength");
      numbers((size_t)3 * (size_t)shape.path_length);
      break;
    case kSetOdometry:
      if (shape.path_length < 1) return fail(p, POCS_E_ORDER, "setOdometry before setPathLength");
      numbers((size_t)3 * (size_t)(shape.path_length - 1));
      break;
    case kSetSeed: {
      while (*rest == ' ' || *rest == '\t') ++rest;
      char* end = nullptr;
      p.seed = strtoull(rest, &end, 0);
      if (end == rest) return fail(p, POCS_E_ARG, "setSeed: integer expected");
      break;
    }
    case kSetFootprint: numbers(4); break;
    case kAddObstacle: numbers(5); break;
    case kUnknown:
      return fail(p, POCS_E_UNKNOWN_COMMAND, "unknown command '%s'", nm);
  }
  return p;
}

}  // namespace pocs_cmd
