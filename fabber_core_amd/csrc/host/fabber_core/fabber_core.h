/* fabber_core.h - command line entry point (the reference's fabber_core.h:19) */
#pragma once

/** Run the `fabber` command line program. Returns 0 if all went well. */
int execute(int argc, char **argv);
