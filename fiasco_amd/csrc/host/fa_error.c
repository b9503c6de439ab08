/*
 *  fa_error.c -- last-error string, verbosity-gated messages, and the two non-public
 *  helpers the reference CLI imports (fiasco_calloc, open_file).
 *
 *  Behaviour follows reference lib/error.c:49-310 (message classes and what each
 *  verbosity level prints), lib/misc.c:51-71 (fiasco_calloc) and lib/bit-io.c:48-146
 *  (open_file search order).  Unlike the reference no longjmp is used: errors are
 *  propagated as return codes and the observable contract (return 0 + message) is kept.
 */
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>
#include "fa_host.h"

static fiasco_verbosity_e g_verbosity = FIASCO_SOME_VERBOSITY; /* lib/error.c:48 */
/* per thread: the writer / upload helpers of a batch run on several threads and publish their
 * message through the task record after the join (the reference is single threaded) */
static __thread char g_error[1024];

void fa_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}

const char *fiasco_get_error_message(void) { return g_error; }
void fiasco_set_verbosity(fiasco_verbosity_e level) { g_verbosity = level; }
fiasco_verbosity_e fiasco_get_verbosity(void) { return g_verbosity; }

static void vmsg(const char *prefix, int newline, const char *fmt, va_list ap)
{
    if (prefix) fputs(prefix, stderr);
    vfprintf(stderr, fmt, ap);
    if (newline) fputc('\n', stderr);
    else fflush(stderr);
}

void fa_warning(const char *fmt, ...)
{
    va_list ap;
    if (g_verbosity == FIASCO_NO_VERBOSITY) return;
    va_start(ap, fmt); vmsg("Warning: ", 1, fmt, ap); va_end(ap);
}

void fa_message(const char *fmt, ...)
{
    va_list ap;
    if (g_verbosity == FIASCO_NO_VERBOSITY) return;
    va_start(ap, fmt); vmsg(NULL, 1, fmt, ap); va_end(ap);
}

void fa_debug(const char *fmt, ...)
{
    va_list ap;
    if (g_verbosity < FIASCO_ULTIMATE_VERBOSITY) return;
    va_start(ap, fmt); vmsg("*** ", 1, fmt, ap); va_end(ap);
}

void fa_progress(const char *fmt, ...)
{
    va_list ap;
    if (g_verbosity == FIASCO_NO_VERBOSITY) return;
    va_start(ap, fmt); vmsg(NULL, 0, fmt, ap); va_end(ap);
}

/* reference lib/misc.c:51-71: zero-size requests and OOM are errors.  The reference
 * longjmps out; a library without a jmp_buf can only report and hand back NULL. */
void *fiasco_calloc(size_t n, size_t size)
{
    void *p;
    if (n == 0 || size == 0) {
        fa_set_error("Can't allocate memory for %d items of size %d", (int) n, (int) size);
        return NULL;
    }
    p = calloc(n, size);
    if (!p) fa_set_error("Out of memory!");
    return p;
}

void fiasco_amd_free(void *p) { free(p); }

#ifndef FIASCO_SHARE
#define FIASCO_SHARE "/usr/local/share/fiasco"
#endif

static FILE *try_dir(const char *dir, const char *filename, const char *mode)
{
    size_t ld = strlen(dir);
    char *full = (char *) malloc(ld + strlen(filename) + 2);
    FILE *fp;
    if (!full) return NULL;
    strcpy(full, dir);
    if (ld == 0 || full[ld - 1] != '/') strcat(full, "/");
    strcat(full, filename);
    fp = fopen(full, mode);
    free(full);
    return fp;
}

/* reference lib/bit-io.c:48-146: "-"/NULL = stdin/stdout; readable file in cwd first;
 * a WRITE name containing '/' is opened as is; otherwise each directory of $env_var
 * (separators " ;:,", default ".") is tried, then FIASCO_SHARE. */
FILE *open_file(const char *filename, const char *env_var, openmode_e mode)
{
    const char *m = mode == READ_ACCESS ? "r" : "w";
    const char *env = NULL;
    char *paths, *tok, *save = NULL;
    FILE *fp = NULL;

    if (!filename || strcmp(filename, "-") == 0)
        return mode == READ_ACCESS ? stdin : stdout;
    if (mode == READ_ACCESS && (fp = fopen(filename, m)))
        return fp;
    if (mode == WRITE_ACCESS && strchr(filename, '/'))
        return fopen(filename, m);
    if (env_var) env = getenv(env_var);
    paths = strdup(env ? env : ".");
    if (!paths) return NULL;
    for (tok = strtok_r(paths, " ;:,", &save); tok && !fp;
         tok = strtok_r(NULL, " ;:,", &save))
        fp = try_dir(tok, filename, m);
    free(paths);
    if (!fp) fp = try_dir(FIASCO_SHARE, filename, m);
    return fp;
}

unsigned char *fa_read_whole_file(const char *name, const char *env_var, size_t *len)
{
    FILE *f = open_file(name, env_var, READ_ACCESS);
    unsigned char *buf = NULL;
    size_t cap = 0, n = 0;
    if (!f) {
        fa_set_error("File `%s': I/O Error - %s.", name ? name : "stdin", strerror(errno));
        return NULL;
    }
    for (;;) {
        size_t r;
        if (n + 65536 > cap) {
            unsigned char *nb;
            cap = cap ? cap * 2 : (1u << 20);
            nb = (unsigned char *) realloc(buf, cap);
            if (!nb) { free(buf); if (f != stdin) fclose(f); fa_set_error("Out of memory!"); return NULL; }
            buf = nb;
        }
        r = fread(buf + n, 1, cap - n, f);
        n += r;
        if (r == 0) break;
    }
    if (f != stdin) fclose(f);
    *len = n;
    return buf;
}
