/* kssd_oracle_cli.c -- TEST INFRASTRUCTURE ONLY: command-line face of the oracle restatement.
 * usage: kssd_oracle_cli -L x.shuf [-A] [-u] [-n M] [-Q q] -o outdir file...   (mirrors `metakssd dist`, -p 1, given file order) */
#include "kssd_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* kssd_oracle_cli set -u|-q [-y] [-o outdir] sketchdir   (mirrors `metakssd set`; -y answers the single-sketch prompt with Y) */
static int set_main(int argc, char **argv) {
  const char *out = "./", *in = NULL, *pan = NULL, *tax = NULL;
  int op = -1, yes = 0;
  for (int i = 0; i < argc; i++) {
    if (!strcmp(argv[i], "-u")) { if (op < 0) op = 0; }
    else if (!strcmp(argv[i], "-i") && i + 1 < argc) { if (op < 0) { op = 3; pan = argv[i + 1]; } i++; }
    else if (!strcmp(argv[i], "-s") && i + 1 < argc) { if (op < 0) { op = 2; pan = argv[i + 1]; } i++; }
    else if (!strcmp(argv[i], "-g") && i + 1 < argc) tax = argv[++i];
    else if (!strcmp(argv[i], "-q")) { if (op < 0) op = 1; }
    else if (!strcmp(argv[i], "-y")) yes = 1;
    else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
    else in = argv[i];
  }
  if (op < 0 && tax && in) { /* cmd_set(): -g is looked at only when no operation was chosen (command_set.c:227-231) */
    int rc = ko_set_group(in, tax, out);
    if (rc) fprintf(stderr, "kssd_oracle_cli set -g: error %d\n", rc);
    return rc ? 1 : 0;
  }
  if (op < 0 || !in) { fprintf(stderr, "usage: kssd_oracle_cli set -u|-q|-i pan|-s pan [-y] [-o outdir] sketchdir\n"); return 2; }
  int rc = op >= 2 ? ko_set_operate(in, pan, out, op == 3) : ko_set_stage(in, out, op, yes);
  if (rc) fprintf(stderr, "kssd_oracle_cli set: error %d\n", rc);
  return rc ? 1 : 0;
}

/* kssd_oracle_cli composite -r refdir -q qrydir [-b] [-o outdir] */
static int composite_main(int argc, char **argv) {
  const char *ref = NULL, *qry = NULL, *out = "./";
  int b = 0;
  for (int i = 0; i < argc; i++) {
    if (!strcmp(argv[i], "-r") && i + 1 < argc) ref = argv[++i];
    else if (!strcmp(argv[i], "-q") && i + 1 < argc) qry = argv[++i];
    else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
    else if (!strcmp(argv[i], "-b")) b = 1;
  }
  if (!ref || !qry) { fprintf(stderr, "usage: kssd_oracle_cli composite -r refdir -q qrydir [-b] [-o outdir]\n"); return 2; }
  int rc = ko_composite(ref, qry, out, b, stdout);
  if (rc) fprintf(stderr, "kssd_oracle_cli composite: error %d\n", rc);
  return rc ? 1 : 0;
}

/* kssd_oracle_cli stage2 [--no-index] -o mcodir codir          (mirrors `metakssd dist -o mcodir codir`)
 * kssd_oracle_cli search [-M m] [-O f] [-N n] [-D d] [--correction c] [--keepskf] [--refco] -r refdir -o outdir qrydir */
static int stage2_main(int argc, char **argv) {
  const char *out = NULL, *in = NULL;
  int dense = 1;
  for (int i = 0; i < argc; i++) {
    if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
    else if (!strcmp(argv[i], "--no-index")) dense = 0;
    else in = argv[i];
  }
  if (!out || !in) { fprintf(stderr, "usage: kssd_oracle_cli stage2 [--no-index] -o mcodir codir\n"); return 2; }
  int rc = ko_stage2(in, out, dense);
  if (rc) fprintf(stderr, "kssd_oracle_cli stage2: error %d\n", rc);
  return rc ? 1 : 0;
}

static int search_main(int argc, char **argv) {
  const char *ref = NULL, *out = NULL, *qry = NULL;
  ko_dist_opts o = {0, 2, 0, 1.0, 0, 0};
  int refco = 0;
  for (int i = 0; i < argc; i++) {
    if (!strcmp(argv[i], "-r") && i + 1 < argc) ref = argv[++i];
    else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
    else if (!strcmp(argv[i], "-M") && i + 1 < argc) o.metric = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-O") && i + 1 < argc) o.outfields = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-N") && i + 1 < argc) o.num_neigb = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-D") && i + 1 < argc) o.dthreshold = atof(argv[++i]);
    else if (!strcmp(argv[i], "--correction") && i + 1 < argc) o.correction = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--keepskf")) o.keep_shared = 1;
    else if (!strcmp(argv[i], "--refco")) refco = 1;
    else qry = argv[i];
  }
  if (!ref || !out || !qry) { fprintf(stderr, "usage: kssd_oracle_cli search [options] -r refdir -o outdir qrydir\n"); return 2; }
  int rc = ko_dist_search(ref, refco, qry, out, &o);
  if (rc) fprintf(stderr, "kssd_oracle_cli search: error %d\n", rc);
  return rc ? 1 : 0;
}

int main(int argc, char **argv) {
  if (argc > 1 && !strcmp(argv[1], "stage2")) return stage2_main(argc - 2, argv + 2);
  if (argc > 1 && !strcmp(argv[1], "search")) return search_main(argc - 2, argv + 2);
  if (argc > 1 && !strcmp(argv[1], "set")) return set_main(argc - 2, argv + 2);
  if (argc > 1 && !strcmp(argv[1], "composite")) return composite_main(argc - 2, argv + 2);
  const char *shuf = NULL, *out = NULL;
  int A = 0, u = 0, nf = 0, Q = 0, M = 1;
  const char *files[4096];
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-L") && i + 1 < argc) shuf = argv[++i];
    else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
    else if (!strcmp(argv[i], "-A")) A = 1;
    else if (!strcmp(argv[i], "-u")) u = 1;
    else if (!strcmp(argv[i], "-Q") && i + 1 < argc) Q = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-n") && i + 1 < argc) { /* command_dist_wrapper.c:169-180 clamps to 1..7 */
      M = atoi(argv[++i]);
      if (M > 7) M = 7;
      if (M < 1) M = 1;
    }
    else if (nf < 4096) files[nf++] = argv[i];
  }
  if (!shuf || !out || nf == 0) {
    fprintf(stderr, "usage: %s -L x.shuf [-A] [-u] -o outdir file...\n", argv[0]);
    return 2;
  }
  int rc = ko_dist_stage1_ex(shuf, A, u, Q, M, out, nf, files);
  if (rc) fprintf(stderr, "kssd_oracle_cli: error %d\n", rc);
  return rc ? 1 : 0;
}
