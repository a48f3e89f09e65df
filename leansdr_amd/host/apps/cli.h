// leansdr_amd/host/apps/cli.h — shared by the generator-side apps: a table-driven option parser and a small owner for the
// scheduler, the device context and the heap-allocated pipes/blocks of one graph.
#ifndef LEANSDR_AMD_APPS_CLI_H
#define LEANSDR_AMD_APPS_CLI_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <vector>

#include "leansdr/framework.h"

namespace cli {

struct option {
  const char *flag;
  const char *arg;    // NULL: a switch; otherwise the placeholder shown by -h
  const char *help;
  std::function<void(const char *)> apply;
};

struct parser {
  const char *summary;
  std::vector<option> options;
  [[noreturn]] void usage(const char *prog, FILE *to, int status) const {
    fprintf(to, "Usage: %s [options]\n%s\n", prog, summary);
    for (const option &o : options) {
      char left[64];
      snprintf(left, sizeof(left), "%s %s", o.flag, o.arg ? o.arg : "");
      fprintf(to, "  %-24s %s\n", left, o.help);
    }
    exit(status);
  }
  void parse(int argc, char **argv) const {
    for (int i = 1; i < argc; ++i) {
      if (!strcmp(argv[i], "-h")) usage(argv[0], stdout, 0);
      const option *hit = NULL;
      for (const option &o : options)
        if (!strcmp(argv[i], o.flag)) hit = &o;
      if (!hit || (hit->arg && i + 1 >= argc)) usage(argv[0], stderr, 1);
      hit->apply(hit->arg ? argv[++i] : NULL);
    }
  }
};

template <typename V>
struct named {
  const char *name;
  V value;
};
template <typename V, size_t N>
bool pick(const char *text, const named<V> (&table)[N], V *out) {
  for (size_t i = 0; i < N; ++i)
    if (!strcmp(text, table[i].name)) { *out = table[i].value; return true; }
  return false;
}

inline float db_to_amplitude(const char *text) { return expf(logf(10) * atof(text) / 20); }

// One flow graph: scheduler + device context; pipes and blocks created through it live until exit.
struct graph {
  leansdr::scheduler sch;
  lsdr_ctx *ctx;
  explicit graph(int device) : ctx(NULL) { leansdr::lsdr_check(lsdr_ctx_create(device, NULL, &ctx), "lsdr_ctx_create"); }
  template <typename T>
  leansdr::pipebuf<T> &host(const char *name, unsigned long items) { return *new leansdr::pipebuf<T>(&sch, name, items); }
  template <typename T>
  leansdr::pipebuf<T> &hbm(const char *name, unsigned long items) { return *new leansdr::pipebuf<T>(&sch, name, items, ctx); }
  void run() {
    sch.run();
    sch.shutdown();
    if (sch.verbose) sch.dump();
    lsdr_ctx_destroy(ctx);
  }
};

}  // namespace cli
#endif
