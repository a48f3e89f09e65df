// tests/host/framework_test.cc — semantics of the host data-flow runtime (leansdr_amd/host/leansdr/framework.h) that graphs rely on,
// checked on host pipes only (no GPU): contiguous reads, compaction on short tail room, pipes without readers, two readers,
// overflow/underflow aborts are not exercised.  Prints "ok" and exits 0, or describes the first failed check.
#include <stdio.h>

#include "leansdr/framework.h"

using namespace leansdr;

#define CHECK(c) do { if (!(c)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

struct counter_source : runnable {   // emits 0 … total−1, as much as fits per pass
  counter_source(scheduler *s, pipebuf<int> &o, int total_) : runnable(s, "source"), out(o), next(0), total(total_) {}
  void run() {
    unsigned long n = out.writable();
    for (unsigned long i = 0; i < n && next < total; ++i) out.write(next++);
  }
  pipewriter<int> out;
  int next, total;
};
struct summing_sink : runnable {     // consumes at most `greed` items per pass
  summing_sink(scheduler *s, pipebuf<int> &i, unsigned long greed_) : runnable(s, "sink"), in(i), greed(greed_), sum(0), seen(0), ordered(true) {}
  void run() {
    unsigned long n = min(in.readable(), greed);
    for (unsigned long i = 0; i < n; ++i) {
      if (in.rd()[i] != (int)seen) ordered = false;
      sum += in.rd()[i];
      ++seen;
    }
    in.read(n);
  }
  pipereader<int> in;
  unsigned long greed;
  long sum, seen;
  bool ordered;
};

int main() {
  {   // compaction when the tail room falls below the writer's minimum; unread data stays contiguous and intact
    scheduler sch;
    pipebuf<int> p(&sch, "p", 16);
    pipewriter<int> w(p, 4);
    pipereader<int> r(p);
    CHECK(w.writable() == 16);
    for (int i = 0; i < 14; ++i) w.write(i);
    CHECK(r.readable() == 14 && r.rd()[13] == 13);
    r.read(12);
    CHECK(w.writable() == 14);          // tail room was 2 < 4 → the 2 unread items moved to the front
    CHECK(r.readable() == 2 && r.rd()[0] == 12 && r.rd()[1] == 13);
    w.write(14);
    CHECK(r.readable() == 3 && r.rd()[2] == 14);
  }
  {   // tail room ≥ min_write: no compaction, the write pointer keeps advancing
    scheduler sch;
    pipebuf<int> p(&sch, "p", 16);
    pipewriter<int> w(p, 2);
    pipereader<int> r(p);
    for (int i = 0; i < 10; ++i) w.write(i);
    r.read(10);
    CHECK(w.writable() == 6);
  }
  {   // a pipe nobody reads never fills up
    scheduler sch;
    pipebuf<int> p(&sch, "unread", 8);
    pipewriter<int> w(p, 8);
    for (int round = 0; round < 5; ++round) {
      CHECK(w.writable() == 8);
      for (int i = 0; i < 8; ++i) w.write(i);
    }
  }
  {   // two readers: the slower one bounds what can be reclaimed
    scheduler sch;
    pipebuf<int> p(&sch, "p", 8);
    pipewriter<int> w(p, 8);
    pipereader<int> fast(p), slow(p);
    for (int i = 0; i < 8; ++i) w.write(i);
    fast.read(8);
    slow.read(3);
    CHECK(w.writable() == 3);           // 5 items still owed to the slow reader
    CHECK(slow.readable() == 5 && slow.rd()[0] == 3 && fast.readable() == 0);
  }
  {   // the scheduler runs to the fixpoint: everything produced is consumed, in order, through a pipe smaller than the stream
    scheduler sch;
    pipebuf<int> p(&sch, "stream", 64);
    counter_source src(&sch, p, 10000);
    summing_sink snk(&sch, p, 7);
    sch.run();
    CHECK(snk.seen == 10000 && snk.ordered && snk.sum == 10000L * 9999 / 2);
  }
  printf("ok\n");
  return 0;
}
