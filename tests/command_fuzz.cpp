// command_fuzz.cpp -- drives the module's text grammar (csrc/pocs_command.hpp: host only) from records on stdin, for
// the fuzz of tests/test_sanitizers.py.  Always built with -fsanitize=address,undefined: the reference's handlers
// overflow (setAlphas with a fifth token, mcsimplugin.cpp:176-184) and run off the end (eight handlers without a
// return, :83-172) exactly here, and this is where a drop-in must not.
//   record := <num_landmarks> ' ' <path_length> ' ' <len> ' ' <len bytes of the command line>
//   reply  := id err nvalues n seed '|' name '|' msg '\n' then nvalues values ("%.17g"), one per line
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#include "../probability-of-collision-for-safe-planning_amd/csrc/pocs_command.hpp"

int main() {
  std::string in;
  char buf[1 << 16];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, stdin)) > 0) in.append(buf, got);
  size_t pos = 0;
  long records = 0;
  while (pos < in.size()) {
    char* end = nullptr;
    const long nl = strtol(in.c_str() + pos, &end, 10);
    const long W = strtol(end, &end, 10);
    const long len = strtol(end, &end, 10);
    pos = (size_t)(end - in.c_str()) + 1;                 // the one space behind <len>
    if (len < 0 || pos + (size_t)len > in.size()) { fprintf(stderr, "bad record %ld\n", records); return 2; }
    // an exactly-sized heap copy, NUL-terminated: a read past the line's end is a heap-buffer-overflow for ASan
    char* line = (char*)malloc((size_t)len + 1);
    for (long i = 0; i < len; ++i) line[i] = in[pos + (size_t)i];
    line[len] = 0;
    pos += (size_t)len;
    const pocs_cmd::Shape shape = {(int)nl, (int)W};
    const pocs_cmd::Parsed p = pocs_cmd::parse(line, shape);
    free(line);
    std::string name = p.name, msg = p.msg;
    for (char& ch : name) if (ch == '\n' || ch == '|') ch = '?';
    for (char& ch : msg) if (ch == '\n' || ch == '|') ch = '?';
    printf("%d %d %zu %lld %llu|%s|%s\n", (int)p.id, p.err, p.v.size(), p.n, p.seed, name.c_str(), msg.c_str());
    for (double v : p.v) printf("%.17g\n", v);
    ++records;
  }
  fprintf(stderr, "%ld records\n", records);
  return 0;
}
