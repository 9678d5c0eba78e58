/* placeholder until the RANSAC/Kabsch oracle lands */
int oracle_ransac_placeholder(void) { return 0; }
